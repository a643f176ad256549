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

// ---- batched form: every agent of a modality in ONE launch chain -----------------------------------------------------
// The chain above is launch-bound (~15 kernels of <= 10 us for a 55 k-point sweep), so n agents cost n chains.  Here the
// agents' clouds are concatenated (host-known offsets), cell keys carry the agent (agent * cells + cell) and ONE chain of
// the same length serves all of them: the scan of the "first point of its cell" flags numbers voxels in (agent,
// first-appearance) order, which is exactly the collated layout of collate_batch_list.
constexpr int VOX_MAX_BATCH = 16;
struct VoxBatch {
    int B;
    int pt_off[VOX_MAX_BATCH + 1];  // points of agent b: [pt_off[b], pt_off[b+1])
};

__device__ __forceinline__ int vox_agent(const VoxBatch& vb, int i) {
    int a = 0;
#pragma unroll 1
    while (a + 1 < vb.B && i >= vb.pt_off[a + 1]) ++a;
    return a;
}

__global__ __launch_bounds__(256) void k_voxb_insert(const float4* __restrict__ pts, VoxBatch vb, VoxGrid g,
                                                    uint32_t cells, uint32_t* __restrict__ tkey,
                                                    uint32_t* __restrict__ tmin, uint32_t mask,
                                                    int* __restrict__ slot_of) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= vb.pt_off[vb.B]) return;
    int cx, cy, cz;
    if (!point_cell(pts[i], g, cx, cy, cz)) { slot_of[i] = -1; return; }
    const uint32_t cell = (uint32_t)vox_agent(vb, i) * cells +
                          (((uint32_t)cz * (uint32_t)g.grid[1] + (uint32_t)cy) * (uint32_t)g.grid[0] + (uint32_t)cx);
    uint32_t slot = hash_u32(cell) & mask;
    for (;;) {
        const uint32_t prev = atomicCAS(&tkey[slot], HASH_EMPTY, cell);
        if (prev == HASH_EMPTY || prev == cell) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&tmin[slot], (uint32_t)i);
    slot_of[i] = (int)slot;
}

// per-agent bases: vbase[b] = voxels found before agent b's first point (scan value there), obase[b] = first output row
// of agent b = sum over earlier agents of min(found, max_voxels); offsets_out[0..B] = obase (collated row offsets)
__global__ void k_voxb_bases(const int* __restrict__ vid_excl, const int* __restrict__ total, VoxBatch vb,
                             int max_voxels, int* __restrict__ vbase, int* __restrict__ obase,
                             int* __restrict__ offsets_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = vb.pt_off[vb.B];
    int acc = 0;
    for (int b = 0; b <= vb.B; ++b) {
        const int v = (b == vb.B || vb.pt_off[b] >= n) ? *total : vid_excl[vb.pt_off[b]];
        vbase[b] = v;
        if (b > 0) acc += min(v - vbase[b - 1], min(max_voxels, vb.pt_off[b] - vb.pt_off[b - 1]));
        obase[b] = acc;
        offsets_out[b] = acc;
    }
}

__global__ __launch_bounds__(256) void k_voxb_assign(const float4* __restrict__ pts, VoxBatch vb, VoxGrid g,
                                                    const int* __restrict__ slot_of,
                                                    const uint32_t* __restrict__ tmin,
                                                    const int* __restrict__ vid_excl, int max_voxels, int sentinel,
                                                    const int* __restrict__ vbase, const int* __restrict__ obase,
                                                    uint32_t* __restrict__ tvid, int* __restrict__ coords) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= vb.pt_off[vb.B]) return;
    const int s = slot_of[i];
    if (s < 0 || tmin[s] != (uint32_t)i) return;
    const int a = vox_agent(vb, i);
    const int vl = vid_excl[i] - vbase[a];  // first-appearance rank inside the agent
    if (vl < max_voxels) {
        const int row = obase[a] + vl;
        tvid[s] = (uint32_t)row;
        int cx, cy, cz;
        point_cell(pts[i], g, cx, cy, cz);
        reinterpret_cast<int4*>(coords)[row] = make_int4(a, cz, cy, cx);
    } else {
        tvid[s] = (uint32_t)sentinel;  // voxels past max_voxels are dropped
    }
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

namespace heal {
struct PointMask {
    float lo[3], hi[3];
    int use_range, use_ego;
};

__global__ __launch_bounds__(256) void k_mask_points(const float4* pts, int n, PointMask m, float4* out) {  // may alias
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    bool keep = true;
    if (m.use_range)  // pcd_utils.py:58-63 (a NaN coordinate fails every comparison: dropped, like numpy)
        keep = p.x > m.lo[0] && p.x < m.hi[0] && p.y > m.lo[1] && p.y < m.hi[1] && p.z > m.lo[2] && p.z < m.hi[2];
    if (m.use_ego) {  // pcd_utils.py:84-86
        const bool body = p.x >= -1.95f && p.x <= 2.95f && p.y >= -1.1f && p.y <= 1.1f;
        keep = keep && !body;
    }
    const float nan = __int_as_float(0x7fc00000);
    out[i] = keep ? p : float4{nan, nan, nan, nan};
}
}  // namespace heal

extern "C" int heal_mask_points(const float* points, int n_points, const float* range_host, int mask_ego, float* out,
                                void* stream) {
    HEAL_REQUIRE(n_points >= 0, "mask_points: bad size");
    if (n_points == 0) return 0;
    HEAL_REQUIRE(points && out, "mask_points: null pointer");
    HEAL_REQUIRE((((uintptr_t)points | (uintptr_t)out) & 15) == 0, "mask_points: points / out must be 16-byte aligned");
    heal::PointMask m;
    m.use_range = range_host != nullptr;
    m.use_ego = mask_ego != 0;
    for (int k = 0; k < 3; ++k) {
        m.lo[k] = range_host ? range_host[k] : 0.f;
        m.hi[k] = range_host ? range_host[3 + k] : 0.f;
    }
    heal::k_mask_points<<<heal::ceil_div(n_points, 256), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(points), n_points, m, reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
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


extern "C" size_t heal_voxelize_batch_workspace(int n_points_total, int n_agents) {
    if (n_points_total < 1) n_points_total = 1;
    Arena a(nullptr, 0);
    VoxWs w;
    carve(a, n_points_total, n_points_total, w);
    return a.off + align_up((size_t)(2 * (VOX_MAX_BATCH + 1)) * sizeof(int)) + 256;
}

extern "C" int heal_voxelize_batch(const float* points, const int32_t* point_offsets_host, int n_agents,
                                   const float* range_host, const float* voxel_size_host, int max_points,
                                   int max_voxels, float* voxels, int32_t* coords, int32_t* num_points,
                                   int32_t* row_offsets, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= VOX_MAX_BATCH, "voxelize_batch: 1..%d agents per call", VOX_MAX_BATCH);
    HEAL_REQUIRE(max_points >= 1 && max_voxels >= 1 && row_offsets != nullptr, "voxelize_batch: bad arguments");
    VoxBatch vb;
    vb.B = n_agents;
    for (int b = 0; b <= n_agents; ++b) {
        vb.pt_off[b] = point_offsets_host[b];
        HEAL_REQUIRE(b == 0 ? vb.pt_off[0] == 0 : vb.pt_off[b] >= vb.pt_off[b - 1], "voxelize_batch: offsets must ascend from 0");
    }
    const int n = vb.pt_off[n_agents];
    if (n == 0) {
        HEAL_HIP(hipMemsetAsync(row_offsets, 0, sizeof(int) * (size_t)(n_agents + 1), s));
        return 0;
    }
    VoxGrid g;
    int64_t cells = 1;
    for (int j = 0; j < 3; ++j) {
        g.rmin[j] = range_host[j];
        g.vsize[j] = voxel_size_host[j];
        double gs = ((double)range_host[3 + j] - (double)range_host[j]) / (double)voxel_size_host[j];
        g.grid[j] = (int)__builtin_rint(gs);
        HEAL_REQUIRE(g.grid[j] >= 1, "voxelize_batch: empty grid on axis %d", j);
        cells *= g.grid[j];
    }
    HEAL_REQUIRE(cells * n_agents < 0xFFFFFFFFll, "voxelize_batch: agents x cells exceeds 32-bit keys");
    int cap = 0;  // rows of the collated outputs: sum of min(n_b, max_voxels)
    for (int b = 0; b < n_agents; ++b) cap += (vb.pt_off[b + 1] - vb.pt_off[b]) < max_voxels ? (vb.pt_off[b + 1] - vb.pt_off[b]) : max_voxels;
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "voxelize_batch: workspace must be 256-B aligned");
    Arena a(ws, ws_bytes);
    VoxWs w;
    HEAL_REQUIRE(carve(a, n, n, w), "voxelize_batch: workspace too small (%zu < %zu)", ws_bytes, a.off);
    int* vbase = a.take<int>(VOX_MAX_BATCH + 1);
    int* obase = a.take<int>(VOX_MAX_BATCH + 1);
    HEAL_REQUIRE(a.ok(), "voxelize_batch: workspace too small");

    const float4* pts = reinterpret_cast<const float4*>(points);
    const int nb = ceil_div(n, 256);
    HEAL_HIP(hipMemsetAsync(w.tkey, 0xFF, (size_t)((char*)w.tvid - (char*)w.tkey), s));
    HEAL_HIP(hipMemsetAsync(w.count, 0, (size_t)(cap + 1) * sizeof(int), s));
    k_voxb_insert<<<nb, 256, 0, s>>>(pts, vb, g, (uint32_t)cells, w.tkey, w.tmin, w.tcap - 1, w.slot_of);
    k_vox_flag<<<nb, 256, 0, s>>>(w.slot_of, w.tmin, n, w.flag);
    if (scan_exclusive(w.flag, w.flag, n, w.total, w.scratch, s)) return 1;
    k_voxb_bases<<<1, 64, 0, s>>>(w.flag, w.total, vb, max_voxels, vbase, obase, row_offsets);
    k_voxb_assign<<<nb, 256, 0, s>>>(pts, vb, g, w.slot_of, w.tmin, w.flag, max_voxels, cap, vbase, obase, w.tvid,
                                     coords);
    k_vox_keys<<<nb, 256, 0, s>>>(w.slot_of, w.tvid, n, cap, w.keys[0], w.vals[0], w.count);
    int res = 0;
    if (radix_sort_pairs(w.keys, w.vals, n, key_bits_for(cap), &res, w.scratch, s)) return 1;
    k_vox_heads<<<nb, 256, 0, s>>>(w.keys[res], n, cap, w.seg_start);
    // rows written: row_offsets[n_agents] (device); n_voxels_out goes to a scratch word
    k_vox_write<<<ceil_div(cap, 4), 256, 0, s>>>(pts, w.vals[res], w.seg_start, w.count, row_offsets + n_agents, cap,
                                                 max_points, reinterpret_cast<float4*>(voxels), num_points, w.total + 1,
                                                 nullptr, nullptr);
    HEAL_LAUNCH_CHECK();
    return 0;
}
