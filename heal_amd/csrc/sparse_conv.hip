// K3 -- submanifold / strided sparse 3-D convolution for the SECOND encoder, plus MeanVFE and the
// sparse -> dense BEV step.
//
// Reference call sites (the arithmetic lives in the third-party spconv, not in the reference tree;
// behaviour restated from SURVEY Appendix A2): opencood/models/sub_modules/sparse_backbone_3d.py:11-30
// (conv + BatchNorm1d(eps 1e-3) + ReLU blocks), :48-91 (the 12 layers of VoxelBackBone8x), :114-130
// (forward), opencood/models/sub_modules/mean_vfe.py:13-31, height_compression.py:10-26 (.dense() and
// the [N, C*D, H, W] view).
//
// MI355X formulation
//   * active sites are kept sorted by their linear coordinate, so a tile of consecutive sites is a
//     spatially coherent neighbourhood;
//   * a hash grid (linear coordinate -> site row) answers neighbour queries; the neighbour table
//     nbr[out_site][tap] is built once per `indice_key` and reused by the layers that share it;
//   * strided layers: candidate output cells are deduplicated through a second hash grid, compacted
//     with a prefix sum and radix-sorted -> deterministic site order;
//   * k_sp_conv: output-stationary gather-GEMM on the fp32 matrix cores.  A block owns 64 output sites
//     (16 per wave).  Per kernel tap the [Cin x Cout] weight slab and the 16 gathered input rows of
//     each wave are staged in LDS (padded rows: conflict-free fragment reads), then
//     v_mfma_f32_16x16x4_f32 accumulates; taps that have no neighbour anywhere in the block are
//     skipped.  BatchNorm scale/shift and ReLU are the epilogue.  fp32 in, fp32 accumulate: the
//     result is a fixed-order fmaf chain per output (deterministic, no atomics).
#include <stdlib.h>
#include "prims.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr uint32_t SP_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t sp_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct SpShape {
    int B, D, H, W;
};

__device__ __forceinline__ uint32_t sp_key(const SpShape& s, int b, int z, int y, int x) {
    // unsigned arithmetic: the cell count is < 2^32 (shape_ok) but exceeds 2^31 from batch 13 on at [41,2048,2048]
    return (((uint32_t)b * (uint32_t)s.D + (uint32_t)z) * (uint32_t)s.H + (uint32_t)y) * (uint32_t)s.W + (uint32_t)x;
}

__device__ __forceinline__ int sp_lookup(const uint32_t* __restrict__ tkey, const int* __restrict__ tval,
                                         uint32_t mask, uint32_t key) {
    uint32_t slot = sp_hash(key) & mask;
    for (;;) {
        const uint32_t k = tkey[slot];
        if (k == key) return tval[slot];
        if (k == SP_EMPTY) return -1;
        slot = (slot + 1) & mask;
    }
}

// Row counts may live on the device (n_dev != NULL): the host then passes the CAPACITY of the buffers and every kernel
// clamps to min(*n_dev, capacity) -- no host round trip between the layers, the whole encoder is graph-capturable.
__device__ __forceinline__ int live_rows(const int* __restrict__ n_dev, int cap) {
    return n_dev ? min(*n_dev, cap) : cap;
}

// ---- MeanVFE ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mean_vfe(const float* __restrict__ voxels,
                                                 const int* __restrict__ num, int M_cap, int P, int F,
                                                 const int* __restrict__ n_dev, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int M = live_rows(n_dev, M_cap);
    if (t >= M * F) return;
    const int m = t / F, f = t - m * F;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += voxels[((size_t)m * P + p) * F + f];  // sum over ALL rows (mean_vfe.py:27)
    const float nrm = fmaxf((float)num[m], 1.0f);
    out[t] = s / nrm;
}

// ---- keys / hash -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sp_keys(const int4* __restrict__ idx, int cap, const int* __restrict__ n_dev,
                                                SpShape s, uint32_t* __restrict__ keys,
                                                uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    vals[i] = (uint32_t)i;
    if (i >= live_rows(n_dev, cap)) { keys[i] = SP_EMPTY; return; }  // padding rows sort last (stable: they come last)
    const int4 c = idx[i];  // (b, z, y, x)
    keys[i] = sp_key(s, c.x, c.y, c.z, c.w);
}

__global__ __launch_bounds__(256) void k_sp_apply_perm(const int4* __restrict__ idx,
                                                      const uint32_t* __restrict__ perm, int cap,
                                                      const int* __restrict__ n_dev,
                                                      int4* __restrict__ idx_sorted, int* __restrict__ perm_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const uint32_t j = perm[i];
    idx_sorted[i] = idx[j];
    perm_out[i] = (int)j;
}

__global__ __launch_bounds__(256) void k_sp_hash_insert(const int4* __restrict__ idx, int cap,
                                                       const int* __restrict__ n_dev, SpShape s,
                                                       uint32_t* __restrict__ tkey, int* __restrict__ tval,
                                                       uint32_t mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const int4 c = idx[i];
    const uint32_t key = sp_key(s, c.x, c.y, c.z, c.w);
    uint32_t slot = sp_hash(key) & mask;
    for (;;) {
        const uint32_t prev = atomicCAS(&tkey[slot], SP_EMPTY, key);
        if (prev == SP_EMPTY || prev == key) break;
        slot = (slot + 1) & mask;
    }
    tval[slot] = i;  // sites are unique
}

struct SpConvGeom {
    int k[3], s[3], p[3];  // (z, y, x)
    SpShape in, out;
};

// neighbour table: nbr[o][tap] = input row feeding output o through tap (kz,ky,kx), or -1.
// input coordinate = o*s - p + tap   (cross-correlation, taps enumerated (kz,ky,kx) row-major)
__global__ __launch_bounds__(256) void k_sp_nbr(const int4* __restrict__ out_idx, int out_cap,
                                               const int* __restrict__ n_dev, SpConvGeom g,
                                               const uint32_t* __restrict__ tkey,
                                               const int* __restrict__ tval, uint32_t mask,
                                               int* __restrict__ nbr) {
    const int K = g.k[0] * g.k[1] * g.k[2];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)live_rows(n_dev, out_cap) * K) return;
    const int o = (int)(t / K), tap = (int)(t - (long long)o * K);
    const int kz = tap / (g.k[1] * g.k[2]), ky = (tap / g.k[2]) % g.k[1], kx = tap % g.k[2];
    const int4 c = out_idx[o];
    const int z = c.y * g.s[0] - g.p[0] + kz;
    const int y = c.z * g.s[1] - g.p[1] + ky;
    const int x = c.w * g.s[2] - g.p[2] + kx;
    int r = -1;
    if (z >= 0 && z < g.in.D && y >= 0 && y < g.in.H && x >= 0 && x < g.in.W)
        r = sp_lookup(tkey, tval, mask, sp_key(g.in, c.x, z, y, x));
    nbr[t] = r;
}

// strided conv: every (input site, tap) proposes the output cell it contributes to
__global__ __launch_bounds__(256) void k_sp_candidates(const int4* __restrict__ in_idx, int in_cap,
                                                      const int* __restrict__ n_dev, SpConvGeom g,
                                                      uint32_t* __restrict__ okey, uint32_t omask) {
    const int K = g.k[0] * g.k[1] * g.k[2];
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)live_rows(n_dev, in_cap) * K) return;
    const int i = (int)(t / K), tap = (int)(t - (long long)i * K);
    const int kz = tap / (g.k[1] * g.k[2]), ky = (tap / g.k[2]) % g.k[1], kx = tap % g.k[2];
    const int4 c = in_idx[i];
    const int nz = c.y + g.p[0] - kz, ny = c.z + g.p[1] - ky, nx = c.w + g.p[2] - kx;
    if (nz < 0 || ny < 0 || nx < 0) return;
    if (nz % g.s[0] || ny % g.s[1] || nx % g.s[2]) return;
    const int oz = nz / g.s[0], oy = ny / g.s[1], ox = nx / g.s[2];
    if (oz >= g.out.D || oy >= g.out.H || ox >= g.out.W) return;
    const uint32_t key = sp_key(g.out, c.x, oz, oy, ox);
    uint32_t slot = sp_hash(key) & omask;
    for (;;) {
        const uint32_t prev = atomicCAS(&okey[slot], SP_EMPTY, key);
        if (prev == SP_EMPTY || prev == key) break;
        slot = (slot + 1) & omask;
    }
}

__global__ __launch_bounds__(256) void k_sp_slot_flags(const uint32_t* __restrict__ okey, int cap,
                                                      int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < cap) flag[i] = okey[i] != SP_EMPTY;
}

__global__ __launch_bounds__(256) void k_sp_compact(const uint32_t* __restrict__ okey, const int* __restrict__ pos,
                                                   int cap, int out_cap, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    const uint32_t k = okey[i];
    if (k == SP_EMPTY) return;
    const int p = pos[i];
    if (p < out_cap) { keys[p] = k; vals[p] = (uint32_t)p; }
}

__global__ __launch_bounds__(256) void k_sp_keys_to_idx(const uint32_t* __restrict__ skeys,
                                                       const int* __restrict__ n_dev, int out_cap, SpShape s,
                                                       int4* __restrict__ out_idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(*n_dev, out_cap);
    if (i >= n) return;
    uint32_t k = skeys[i];
    const int x = k % s.W; k /= s.W;
    const int y = k % s.H; k /= s.H;
    const int z = k % s.D; k /= s.D;
    out_idx[i] = make_int4((int)k, z, y, x);
}

// ---- gather-GEMM on the fp32 matrix cores -----------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;

// 32 output sites (two 16-row MFMA tiles) per block, the K taps SPLIT over the block's 4 waves.
//   * the kernel is gather-latency-bound (per tap: neighbour ids -> row gather -> LDS -> MFMA, a serial chain); with
//     split-K each wave walks a quarter of the chain on its own LDS slice, no barrier until the final reduction, and a
//     layer with 15 k active sites is ~470 blocks x 4 independent waves instead of 235 blocks that serialise 27 taps
//     behind four block barriers each (64->64 layer: 144 -> see DESIGN.md);
//   * taps with no neighbour among the 32 sites are skipped (wave-uniform ballot);
//   * A (gathered neighbour rows): coalesced row loads into the wave's LDS slice (64/CIN rows per instruction for
//     narrow layers), fragments read back conflict-free (row stride CIN+4 floats);
//   * B (the tap's [CIN,COUT] weight slice): read straight from the reference layout in fragment order -- lane (k, n)
//     reads W[tap][4kc + k][16nb + n], four 64-B runs per load, L2-resident -- no LDS staging, no re-layout;
//   * fixed summation order (wave w: taps w, w+4, ...; then ((w0 + w1) + w2) + w3): deterministic.
template <int CIN, int COUT, int NW /*waves per block = tap split*/, int MT /*16-row tiles per block*/>
__global__ __launch_bounds__(64 * NW) void k_sp_conv(const float* __restrict__ feat_in,
                                                const int* __restrict__ nbr, int out_cap,
                                                const int* __restrict__ n_dev, int K,
                                                const float* __restrict__ weight /*[K][CIN][COUT]*/,
                                                const float* __restrict__ scale,
                                                const float* __restrict__ shift, int relu,
                                                float* __restrict__ feat_out /*[n_out][COUT]*/) {
    const int n_out = live_rows(n_dev, out_cap);
    if ((int)blockIdx.x * 16 * MT >= n_out) return;  // blocks beyond the live rows (capacity-sized grid)
    constexpr int KC = (CIN + 3) / 4;        // k-steps of 4 input channels
    constexpr int CINP = KC * 4;             // input channels padded to a multiple of 4
    constexpr int NC = COUT / 16;            // 16-wide output-channel blocks
    constexpr int ASTR = CINP + 4;           // LDS row stride: (m*ASTR + k) mod 64 distinct for m < 16, k < 4
    constexpr int MS = 16 * MT;              // sites per block
    constexpr int LPR = CINP < 64 ? CINP : 64;  // lanes per gathered row
    constexpr int RPI = 64 / LPR;            // rows per gather instruction
    constexpr int ACCF = MT * NC * 4;        // accumulator floats per lane
    constexpr int SA = MS * ASTR;            // gather slice per wave
    constexpr int SMEM = (NW * SA > NW * 64 * ACCF) ? NW * SA : NW * 64 * ACCF;
    __shared__ float smem[SMEM];

    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const int site0 = blockIdx.x * MS;
    float* sA = smem + wave * SA;
    f32x4 acc[MT][NC];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int grow = l / LPR, gch = l - grow * LPR;  // gather role of this lane

    for (int tap = wave; tap < K; tap += NW) {
        int src = -1;
        if (l < MS && site0 + l < n_out) src = nbr[(size_t)(site0 + l) * K + tap];
        const unsigned long long have = __ballot(src >= 0);
        if (!have) continue;  // wave-uniform: none of the 32 sites has a neighbour through this tap
        // B fragments of this tap (independent loads, in flight during the gather)
        float bfr[KC][NC];
        const float* wt = weight + (size_t)tap * CIN * COUT;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int ci = kc * 4 + lk;
#pragma unroll
            for (int n = 0; n < NC; ++n) bfr[kc][n] = ci < CIN ? wt[(size_t)ci * COUT + n * 16 + ln] : 0.f;
        }
        // gather the neighbour rows (zero rows where there is none)
#pragma unroll
        for (int r0 = 0; r0 < MS; r0 += RPI) {
            const int r = r0 + grow;
            const int j = __shfl(src, r, 64);
            float v = 0.f;
            if (j >= 0 && gch < CIN) v = feat_in[(size_t)j * CIN + gch];
            if (gch < CINP) sA[r * ASTR + gch] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool live[MT];  // skip an all-zero 16-row tile
#pragma unroll
        for (int m = 0; m < MT; ++m) live[m] = ((have >> (16 * m)) & 0xFFFFull) != 0;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = sA[(16 * m + ln) * ASTR + kc * 4 + lk];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (live[m]) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bfr[kc][n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // before this wave's next tap overwrites its slice
    }
    // reduce the 4 partial accumulators in a fixed order through LDS (the gather slices are free now)
    __syncthreads();
    float* part = smem + ((size_t)wave * 64 + l) * ACCF;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(m * NC + n) * 4 + r] = acc[m][n][r];
    __syncthreads();
    // epilogue: wave w finishes the output-channel blocks n = w, w+4, ...; BatchNorm1d (eval) + ReLU on the active
    // sites; C/D layout: col = l&15, row = (l>>4)*4 + reg
    for (int n = wave; n < NC; n += NW) {
        const int co = n * 16 + ln;
        const float sc = scale[co], sh = shift[co];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (m * NC + n) * 4 + r;
                float sum = smem[(size_t)l * ACCF + e];
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += smem[((size_t)w * 64 + l) * ACCF + e];
                const int site = site0 + m * 16 + lk * 4 + r;
                if (site < n_out) {
                    float v = fmaf(sum, sc, sh);
                    if (relu) v = fmaxf(v, 0.f);
                    feat_out[(size_t)site * COUT + co] = v;
                }
            }
    }
}

// ---- sparse -> dense BEV ([B, C*D, H, W], channel = c*D + z : height_compression.py:21-23) ------------
__global__ __launch_bounds__(256) void k_sp_fill_map(const int4* __restrict__ idx, int cap,
                                                    const int* __restrict__ n_dev, SpShape s,
                                                    int* __restrict__ cell_map /*[B][D][H*W]*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const int4 c = idx[i];
    cell_map[(size_t)((c.x * s.D + c.y) * s.H + c.z) * s.W + c.w] = i;
}

__global__ __launch_bounds__(256) void k_sp_dense(const int4* __restrict__ cell_map4,
                                                 const float* __restrict__ rows, int cells4, int C, int D,
                                                 float4* __restrict__ out4) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= cells4) return;
    const int bz = blockIdx.z;  // b*D + z
    const int b = bz / D, z = bz - b * D;
    const int c0 = blockIdx.y * 16;
    const int4 id = cell_map4[(size_t)bz * cells4 + t];
    const bool empty = (id.x & id.y & id.z & id.w) < 0;
#pragma unroll 4
    for (int c = c0; c < c0 + 16 && c < C; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!empty) {
            v.x = id.x >= 0 ? rows[(size_t)id.x * C + c] : 0.f;
            v.y = id.y >= 0 ? rows[(size_t)id.y * C + c] : 0.f;
            v.z = id.z >= 0 ? rows[(size_t)id.z * C + c] : 0.f;
            v.w = id.w >= 0 ? rows[(size_t)id.w * C + c] : 0.f;
        }
        out4[((size_t)(b * C + c) * D + z) * cells4 + t] = v;
    }
}

static uint32_t pow2_cap(int64_t n) {
    uint32_t c = 1024;
    while ((int64_t)c < 2 * (n < 1 ? 1 : n)) c <<= 1;
    return c;
}

static bool shape_ok(const int* shape, int batch, SpShape& s) {
    s.B = batch; s.D = shape[0]; s.H = shape[1]; s.W = shape[2];
    const int64_t cells = (int64_t)batch * shape[0] * shape[1] * shape[2];
    return batch >= 1 && shape[0] >= 1 && shape[1] >= 1 && shape[2] >= 1 && cells < 0xFFFFFFFFll;
}

}  // namespace heal

using namespace heal;

extern "C" int heal_mean_vfe(const float* voxels, const int32_t* num_points, int n_voxels, int max_points,
                             int n_feat, float* out, const int32_t* n_dev, void* stream) {
    if (n_voxels <= 0) return 0;
    k_mean_vfe<<<ceil_div(n_voxels * n_feat, 256), 256, 0, (hipStream_t)stream>>>(voxels, num_points, n_voxels,
                                                                                 max_points, n_feat, n_dev, out);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_sp_sort_workspace(int n) {
    if (n < 1) n = 1;
    return align_up((size_t)n * 4) * 4 + align_up(sort_scratch_words(n) * 4) + 256;
}

// Sort the active sites by linear coordinate: sorted_indices[i] = indices[perm[i]].
extern "C" int heal_sp_sort_sites(const int32_t* indices, int n, const int32_t* shape_host, int batch,
                                  int32_t* sorted_indices, int32_t* perm, void* ws, size_t ws_bytes,
                                  const int32_t* n_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpShape sh;
    HEAL_REQUIRE(shape_ok(shape_host, batch, sh), "sp_sort_sites: bad shape");
    if (n <= 0) return 0;
    Arena a(ws, ws_bytes);
    uint32_t* keys[2]; uint32_t* vals[2];
    for (int k = 0; k < 2; ++k) { keys[k] = a.take<uint32_t>(n); vals[k] = a.take<uint32_t>(n); }
    int* scratch = a.take<int>(sort_scratch_words(n));
    HEAL_REQUIRE(a.ok(), "sp_sort_sites: workspace too small");
    const int4* idx = reinterpret_cast<const int4*>(indices);
    k_sp_keys<<<ceil_div(n, 256), 256, 0, s>>>(idx, n, n_dev, sh, keys[0], vals[0]);
    const uint64_t cells = (uint64_t)batch * sh.D * sh.H * sh.W;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < cells) ++bits;
    int res = 0;
    if (radix_sort_pairs(keys, vals, n, bits, &res, scratch, s)) return 1;
    k_sp_apply_perm<<<ceil_div(n, 256), 256, 0, s>>>(idx, vals[res], n, n_dev, reinterpret_cast<int4*>(sorted_indices),
                                                     perm);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_sp_table_capacity(int n) { return pow2_cap(n); }

// Build the hash grid of a site set: table_keys[cap] u32, table_vals[cap] i32 (cap = heal_sp_table_capacity(n)).
extern "C" int heal_sp_hash_build(const int32_t* indices, int n, const int32_t* shape_host, int batch,
                                  uint32_t* table_keys, int32_t* table_vals, size_t table_cap, const int32_t* n_dev,
                                  void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpShape sh;
    HEAL_REQUIRE(shape_ok(shape_host, batch, sh), "sp_hash_build: bad shape");
    HEAL_REQUIRE(table_cap >= pow2_cap(n) && (table_cap & (table_cap - 1)) == 0, "sp_hash_build: bad table capacity");
    HEAL_HIP(hipMemsetAsync(table_keys, 0xFF, table_cap * sizeof(uint32_t), s));
    if (n > 0) {
        k_sp_hash_insert<<<ceil_div(n, 256), 256, 0, s>>>(reinterpret_cast<const int4*>(indices), n, n_dev, sh,
                                                          table_keys, table_vals, (uint32_t)table_cap - 1);
        HEAL_LAUNCH_CHECK();
    }
    return 0;
}

static int fill_geom(SpConvGeom& g, const int32_t* ksize, const int32_t* stride, const int32_t* padding,
                     const int32_t* in_shape, const int32_t* out_shape, int batch) {
    for (int k = 0; k < 3; ++k) { g.k[k] = ksize[k]; g.s[k] = stride[k]; g.p[k] = padding[k]; }
    HEAL_REQUIRE(shape_ok(in_shape, batch, g.in) && shape_ok(out_shape, batch, g.out), "spconv: bad shape");
    HEAL_REQUIRE(g.k[0] >= 1 && g.k[1] >= 1 && g.k[2] >= 1 && g.k[0] * g.k[1] * g.k[2] <= 27 && g.s[0] >= 1 &&
                 g.s[1] >= 1 && g.s[2] >= 1, "spconv: unsupported kernel / stride");
    return 0;
}

// nbr[n_out][K] for output sites `out_indices` reading the input site set behind (table_keys, table_vals).
extern "C" int heal_sp_neighbors(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                                 const int32_t* stride_host, const int32_t* padding_host,
                                 const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                 const uint32_t* table_keys, const int32_t* table_vals, size_t table_cap,
                                 int32_t* nbr, const int32_t* n_out_dev, void* stream) {
    SpConvGeom g;
    if (fill_geom(g, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch)) return 1;
    if (n_out <= 0) return 0;
    const int K = g.k[0] * g.k[1] * g.k[2];
    const long long total = (long long)n_out * K;
    k_sp_nbr<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const int4*>(out_indices), n_out, n_out_dev, g, table_keys, table_vals, (uint32_t)table_cap - 1,
        nbr);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_sp_out_sites_workspace(int n_in, int kernel_volume) {
    if (n_in < 1) n_in = 1;
    const uint32_t cap = pow2_cap((int64_t)n_in * (kernel_volume < 8 ? kernel_volume : 8));
    size_t b = align_up((size_t)cap * 4) * 2;                 // okey, flag/pos
    b += align_up((size_t)cap * 4) * 4;                       // keys[2], vals[2] (<= cap entries)
    b += align_up(sort_scratch_words(cap) * 4) + align_up(scan_scratch_words(cap) * 4) + 512;
    return b;
}

// Active output sites of a strided sparse convolution, sorted by linear coordinate.
// out_indices [out_cap,4]; n_out [1] device.
extern "C" int heal_sp_out_sites(const int32_t* in_indices, int n_in, const int32_t* ksize_host,
                                 const int32_t* stride_host, const int32_t* padding_host,
                                 const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                 int32_t* out_indices, int out_cap, int32_t* n_out, void* ws, size_t ws_bytes,
                                 const int32_t* n_in_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpConvGeom g;
    if (fill_geom(g, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch)) return 1;
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "sp_out_sites: workspace must be 256-B aligned");
    if (n_in <= 0) { HEAL_HIP(hipMemsetAsync(n_out, 0, sizeof(int), s)); return 0; }
    const int K = g.k[0] * g.k[1] * g.k[2];
    const uint32_t cap = pow2_cap((int64_t)n_in * (K < 8 ? K : 8));
    Arena a(ws, ws_bytes);
    uint32_t* okey = a.take<uint32_t>(cap);
    int* flag = a.take<int>(cap);
    uint32_t* keys[2]; uint32_t* vals[2];
    for (int k = 0; k < 2; ++k) { keys[k] = a.take<uint32_t>(cap); vals[k] = a.take<uint32_t>(cap); }
    int* sscratch = a.take<int>(sort_scratch_words(cap));
    int* cscratch = a.take<int>(scan_scratch_words(cap));
    HEAL_REQUIRE(a.ok(), "sp_out_sites: workspace too small (%zu < %zu)", ws_bytes, a.off);
    HEAL_HIP(hipMemsetAsync(okey, 0xFF, (size_t)cap * 4, s));
    const long long total = (long long)n_in * K;
    k_sp_candidates<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(reinterpret_cast<const int4*>(in_indices), n_in,
                                                                    n_in_dev, g, okey, cap - 1);
    k_sp_slot_flags<<<ceil_div((int)cap, 256), 256, 0, s>>>(okey, (int)cap, flag);
    if (scan_exclusive(flag, flag, (int)cap, n_out, cscratch, s)) return 1;
    // unique keys -> compact; pad the tail with 0xFFFFFFFF so that a fixed-size sort puts them last
    HEAL_HIP(hipMemsetAsync(keys[0], 0xFF, (size_t)out_cap * 4, s));
    HEAL_REQUIRE((uint32_t)out_cap <= cap, "sp_out_sites: out_cap larger than the candidate table");
    k_sp_compact<<<ceil_div((int)cap, 256), 256, 0, s>>>(okey, flag, (int)cap, out_cap, keys[0], vals[0]);
    // sort on the bits a real key can occupy; the 0xFF.. padding has all of them set and, the sort
    // being stable, stays behind any real key
    const uint64_t ocells = (uint64_t)batch * g.out.D * g.out.H * g.out.W;
    int bits = 1;
    while (bits < 32 && (1ull << bits) < ocells) ++bits;
    int res = 0;
    if (radix_sort_pairs(keys, vals, out_cap, bits, &res, sscratch, s)) return 1;
    k_sp_keys_to_idx<<<ceil_div(out_cap, 256), 256, 0, s>>>(keys[res], n_out, out_cap, g.out,
                                                            reinterpret_cast<int4*>(out_indices));
    HEAL_LAUNCH_CHECK();
    return 0;
}

// out[o] = act( BN( sum_tap W[tap]^T in[nbr[o][tap]] ) ); weight [K][Cin][Cout].
extern "C" int heal_sp_conv(const float* feat_in, const int32_t* nbr, int n_out, int kernel_volume, int c_in,
                            int c_out, const float* weight, const float* bn_scale, const float* bn_shift,
                            int relu, float* feat_out, const int32_t* n_out_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n_out <= 0) return 0;
    // Block shape sweep on MI355X (SECOND encoder, 43 k voxels, 64->64 layers): 32 sites x 4 tap-waves 97 us;
    // 16 sites x 4: 190 (weight fragments re-read twice as often); 64 sites x 4: 140 (occupancy 2); 32 sites x 8: 106.
    const int blocks = ceil_div(n_out, 32);
#define HEAL_SP_CASE(CI, CO)                                                                              \
    if (c_in == CI && c_out == CO) {                                                                      \
        k_sp_conv<CI, CO, 4, 2><<<blocks, 256, 0, s>>>(feat_in, nbr, n_out, n_out_dev, kernel_volume, weight,    \
                                                       bn_scale, bn_shift, relu, feat_out);               \
        HEAL_LAUNCH_CHECK();                                                                              \
        return 0;                                                                                         \
    }
    HEAL_SP_CASE(4, 16) HEAL_SP_CASE(16, 16) HEAL_SP_CASE(16, 32) HEAL_SP_CASE(32, 32) HEAL_SP_CASE(32, 64)
    HEAL_SP_CASE(64, 64) HEAL_SP_CASE(64, 128) HEAL_SP_CASE(8, 16) HEAL_SP_CASE(64, 16)
#undef HEAL_SP_CASE
    return set_error("sp_conv: channel combination %d -> %d is not instantiated", c_in, c_out);
}

extern "C" size_t heal_sp_to_bev_workspace(int batch, int D, int H, int W) {
    return align_up((size_t)batch * D * H * W * 4) + 256;
}

// Sparse tensor -> dense [B, C*D, H, W] (channel = c*D + z), every element written.
extern "C" int heal_sp_to_bev(const float* features, const int32_t* indices, int n, int channels,
                              const int32_t* shape_host, int batch, float* out, void* ws, size_t ws_bytes,
                              const int32_t* n_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpShape sh;
    HEAL_REQUIRE(shape_ok(shape_host, batch, sh), "sp_to_bev: bad shape");
    HEAL_REQUIRE((sh.H * sh.W) % 4 == 0, "sp_to_bev: H*W must be a multiple of 4");
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "sp_to_bev: workspace must be 256-B aligned");
    Arena a(ws, ws_bytes);
    const size_t cells = (size_t)batch * sh.D * sh.H * sh.W;
    int* cell_map = a.take<int>(cells);
    HEAL_REQUIRE(a.ok(), "sp_to_bev: workspace too small");
    HEAL_HIP(hipMemsetAsync(cell_map, 0xFF, cells * 4, s));
    if (n > 0)
        k_sp_fill_map<<<ceil_div(n, 256), 256, 0, s>>>(reinterpret_cast<const int4*>(indices), n, n_dev, sh, cell_map);
    const int cells4 = sh.H * sh.W / 4;
    dim3 grid(ceil_div(cells4, 256), ceil_div(channels, 16), batch * sh.D);
    k_sp_dense<<<grid, 256, 0, s>>>(reinterpret_cast<const int4*>(cell_map), features, cells4, channels, sh.D,
                                    reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
}
