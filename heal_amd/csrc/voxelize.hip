// K1 -- deterministic hard voxelisation on gfx950.
//
// Semantics restated from spconv's CPU point->voxel generator as called at
// opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:46-68:
// scan points in input order; cell = floor((p - range_min) / voxel_size) per axis in fp32, drop if
// outside [0,grid); a new voxel id is handed out on the first touch of a cell (ids in order of
// first appearance, none beyond max_voxels); a point is appended to its voxel while the voxel holds
// fewer than max_points.
//
// GPU formulation (order-independent, so bit-identical to the sequential scan):
//   1. hash-grid insert: cell -> min point index          (atomicCAS claim + atomicMin)
//   2. flag[i] = "i is the first point of its cell"; voxel id = exclusive prefix sum of flags
//      (ballot/popcount inside the block scan)
//   3. every point looks up its voxel id; key = id (or a sentinel for dropped points)
//   4. stable radix sort of (voxel id, point index): points of a voxel become contiguous, still in
//      input order -> slot = position - segment start
//   5. one 64-lane wave per voxel writes the whole [P,4] row block (points, then zero padding):
//      coalesced 16 B/lane stores and no memset of the output.
#include "prims.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct VoxGrid {
    float rmin[3];
    float vsize[3];
    int grid[3];  // x,y,z
};

__device__ __forceinline__ bool point_cell(const float4 p, const VoxGrid& g, int& cx, int& cy, int& cz) {
    // fp32, one subtraction and one IEEE division per axis, then floor (no contraction: the
    // library is built with -ffp-contract=off)
    const float fx = floorf((p.x - g.rmin[0]) / g.vsize[0]);
    const float fy = floorf((p.y - g.rmin[1]) / g.vsize[1]);
    const float fz = floorf((p.z - g.rmin[2]) / g.vsize[2]);
    // NaN coordinates fail every comparison below and are dropped
    if (!(fx >= 0.f && fx < (float)g.grid[0])) return false;
    if (!(fy >= 0.f && fy < (float)g.grid[1])) return false;
    if (!(fz >= 0.f && fz < (float)g.grid[2])) return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return true;
}

// 1. insert: table_key[slot] = cell, table_min[slot] = min point index; slot_of[i] = slot or -1
__global__ __launch_bounds__(256) void k_vox_insert(const float4* __restrict__ pts, int n, VoxGrid g,
                                                   uint32_t* __restrict__ tkey,
                                                   uint32_t* __restrict__ tmin, uint32_t mask,
                                                   int* __restrict__ slot_of) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    if (!point_cell(pts[i], g, cx, cy, cz)) { slot_of[i] = -1; return; }
    const uint32_t cell = ((uint32_t)cz * (uint32_t)g.grid[1] + (uint32_t)cy) * (uint32_t)g.grid[0] + (uint32_t)cx;
    uint32_t slot = hash_u32(cell) & mask;
    for (;;) {
        const uint32_t prev = atomicCAS(&tkey[slot], HASH_EMPTY, cell);
        if (prev == HASH_EMPTY || prev == cell) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&tmin[slot], (uint32_t)i);
    slot_of[i] = (int)slot;
}

// 2a. flags -> per-tile counts (tile = SCAN_TILE points)
__global__ __launch_bounds__(256) void k_vox_flag(const int* __restrict__ slot_of,
                                                 const uint32_t* __restrict__ tmin, int n,
                                                 int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    flag[i] = (s >= 0 && tmin[s] == (uint32_t)i) ? 1 : 0;
}

// 2b. first points publish their voxel id into the table and write the voxel's coordinates
__global__ __launch_bounds__(256) void k_vox_assign(const float4* __restrict__ pts, int n, VoxGrid g,
                                                   const int* __restrict__ slot_of,
                                                   const uint32_t* __restrict__ tmin,
                                                   const int* __restrict__ vid_excl, int cap,
                                                   int batch_idx, const int* __restrict__ row_offset,
                                                   uint32_t* __restrict__ tvid, int* __restrict__ coords) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int base = row_offset ? *row_offset : 0;  // rows of earlier agents in a shared (collated) buffer
    const int s = slot_of[i];
    if (s < 0 || tmin[s] != (uint32_t)i) return;
    const int vid = vid_excl[i];
    tvid[s] = (uint32_t)vid;  // ids >= cap mark voxels past max_voxels
    if (vid < cap) {
        int cx, cy, cz;
        point_cell(pts[i], g, cx, cy, cz);
        reinterpret_cast<int4*>(coords)[base + vid] = make_int4(batch_idx, cz, cy, cx);
    }
}

// 3. sort keys: voxel id, or `cap` (sorts last) for dropped points; per-voxel point counts
__global__ __launch_bounds__(256) void k_vox_keys(const int* __restrict__ slot_of,
                                                 const uint32_t* __restrict__ tvid, int n, int cap,
                                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 int* __restrict__ count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    uint32_t k = (uint32_t)cap;
    if (s >= 0) {
        const uint32_t v = tvid[s];
        if (v < (uint32_t)cap) { k = v; atomicAdd(&count[v], 1); }
    }
    keys[i] = k;
    vals[i] = (uint32_t)i;
}

// 4b. segment heads in the sorted order
__global__ __launch_bounds__(256) void k_vox_heads(const uint32_t* __restrict__ skeys, int n, int cap,
                                                  int* __restrict__ seg_start) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t k = skeys[j];
    if (k >= (uint32_t)cap) return;
    if (j == 0 || skeys[j - 1] != k) seg_start[k] = j;
}

// 5. one wave per voxel: rows [0,min(cnt,P)) = points in input order, the rest zeros
__global__ __launch_bounds__(256) void k_vox_write(const float4* __restrict__ pts,
                                                  const uint32_t* __restrict__ svals,
                                                  const int* __restrict__ seg_start,
                                                  const int* __restrict__ count,
                                                  const int* __restrict__ total_voxels, int cap, int P,
                                                  float4* __restrict__ voxels, int* __restrict__ num_points,
                                                  int* __restrict__ n_voxels_out,
                                                  const int* __restrict__ row_offset,
                                                  int* __restrict__ row_offset_next) {
    const int M = min(*total_voxels, cap);
    const int base = row_offset ? *row_offset : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *n_voxels_out = M;
        if (row_offset_next) *row_offset_next = base + M;
    }
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= M) return;
    const int l = threadIdx.x & 63;
    const int c = min(count[v], P);
    const int st = seg_start[v];
    for (int p = l; p < P; p += 64) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < c) val = pts[svals[st + p]];
        voxels[(size_t)(base + v) * P + p] = val;
    }
    if (l == 0) num_points[base + v] = c;
}

static int key_bits_for(int cap) {
    int b = 1;
    while ((1u << b) <= (uint32_t)cap) ++b;  // need to represent `cap` itself (the sentinel)
    return b;
}

struct VoxWs {
    uint32_t *tkey, *tmin, *tvid;
    int *slot_of, *flag, *count, *seg_start, *total;
    uint32_t *keys[2], *vals[2];
    int* scratch;
    uint32_t tcap;
};

static uint32_t table_cap(int n) {
    uint32_t c = 1024;
    while (c < 2u * (uint32_t)(n < 1 ? 1 : n)) c <<= 1;
    return c;
}

static bool carve(Arena& a, int n, int cap, VoxWs& w) {
    w.tcap = table_cap(n);
    // tkey | tmin are contiguous so one memset(0xFF) initialises both
    w.tkey = a.take<uint32_t>(w.tcap);
    w.tmin = a.take<uint32_t>(w.tcap);
    w.tvid = a.take<uint32_t>(w.tcap);
    w.slot_of = a.take<int>(n);
    w.flag = a.take<int>(n);
    w.count = a.take<int>(cap + 1);
    w.seg_start = a.take<int>(cap + 1);
    w.total = a.take<int>(64);
    for (int k = 0; k < 2; ++k) { w.keys[k] = a.take<uint32_t>(n); w.vals[k] = a.take<uint32_t>(n); }
    size_t sw = sort_scratch_words(n);
    size_t cw = scan_scratch_words(n);
    w.scratch = a.take<int>(sw > cw ? sw : cw);
    return a.ok();
}

}  // namespace heal

using namespace heal;

extern "C" size_t heal_voxelize_workspace(int n_points, int max_voxels) {
    if (n_points < 1) n_points = 1;
    int cap = n_points < max_voxels ? n_points : max_voxels;
    if (cap < 1) cap = 1;
    Arena a(nullptr, 0);
    VoxWs w;
    carve(a, n_points, cap, w);
    return a.off + 256;
}

extern "C" int heal_voxelize(const float* points, int n_points, const float* range_host,
                             const float* voxel_size_host, int max_points, int max_voxels,
                             int batch_idx, float* voxels, int32_t* coords, int32_t* num_points,
                             int32_t* n_voxels, const int32_t* row_offset, int32_t* row_offset_next, void* ws,
                             size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_points >= 0 && max_points >= 1 && max_voxels >= 1, "voxelize: bad sizes");
    HEAL_REQUIRE(n_voxels != nullptr, "voxelize: n_voxels is NULL");
    if (n_points == 0) {
        HEAL_HIP(hipMemsetAsync(n_voxels, 0, sizeof(int), s));
        if (row_offset_next) {
            if (row_offset) HEAL_HIP(hipMemcpyAsync(row_offset_next, row_offset, sizeof(int), hipMemcpyDeviceToDevice, s));
            else HEAL_HIP(hipMemsetAsync(row_offset_next, 0, sizeof(int), s));
        }
        return 0;
    }
    VoxGrid g;
    int64_t cells = 1;
    for (int j = 0; j < 3; ++j) {
        g.rmin[j] = range_host[j];
        g.vsize[j] = voxel_size_host[j];
        // grid = round((max-min)/size), computed like numpy does on the python floats (fp64)
        double gs = ((double)range_host[3 + j] - (double)range_host[j]) / (double)voxel_size_host[j];
        g.grid[j] = (int)__builtin_rint(gs);
        HEAL_REQUIRE(g.grid[j] >= 1, "voxelize: empty grid on axis %d", j);
        cells *= g.grid[j];
    }
    HEAL_REQUIRE(cells < 0xFFFFFFFFll, "voxelize: grid has too many cells (%lld)", (long long)cells);
    const int cap = n_points < max_voxels ? n_points : max_voxels;
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "voxelize: workspace must be 256-B aligned");
    Arena a(ws, ws_bytes);
    VoxWs w;
    HEAL_REQUIRE(carve(a, n_points, cap, w), "voxelize: workspace too small (%zu < %zu)", ws_bytes, a.off);

    const float4* pts = reinterpret_cast<const float4*>(points);
    const int nb = ceil_div(n_points, 256);
    // tkey and tmin: 0xFF.. = EMPTY / +inf ; counts: 0
    HEAL_HIP(hipMemsetAsync(w.tkey, 0xFF, (size_t)((char*)w.tvid - (char*)w.tkey), s));
    HEAL_HIP(hipMemsetAsync(w.count, 0, (size_t)(cap + 1) * sizeof(int), s));
    k_vox_insert<<<nb, 256, 0, s>>>(pts, n_points, g, w.tkey, w.tmin, w.tcap - 1, w.slot_of);
    k_vox_flag<<<nb, 256, 0, s>>>(w.slot_of, w.tmin, n_points, w.flag);
    if (scan_exclusive(w.flag, w.flag, n_points, w.total, w.scratch, s)) return 1;
    k_vox_assign<<<nb, 256, 0, s>>>(pts, n_points, g, w.slot_of, w.tmin, w.flag, cap, batch_idx, row_offset,
                                    w.tvid, coords);
    k_vox_keys<<<nb, 256, 0, s>>>(w.slot_of, w.tvid, n_points, cap, w.keys[0], w.vals[0], w.count);
    int res = 0;
    if (radix_sort_pairs(w.keys, w.vals, n_points, key_bits_for(cap), &res, w.scratch, s)) return 1;
    k_vox_heads<<<nb, 256, 0, s>>>(w.keys[res], n_points, cap, w.seg_start);
    k_vox_write<<<ceil_div(cap, 4), 256, 0, s>>>(pts, w.vals[res], w.seg_start, w.count, w.total, cap,
                                                 max_points, reinterpret_cast<float4*>(voxels),
                                                 num_points, n_voxels, row_offset, row_offset_next);
    HEAL_LAUNCH_CHECK();
    return 0;
}
