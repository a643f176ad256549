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

// out[i] = src[perm[i]] for rows of C floats (16-B pieces when C % 4 == 0): the features of a site set re-ordered by the
// permutation heal_sp_sort_sites returns.
template <typename V>
__global__ __launch_bounds__(256) void k_sp_gather_rows(const V* __restrict__ src, const int* __restrict__ perm, int cap,
                                                       const int* __restrict__ n_dev, int pieces, V* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int i = (int)(e / pieces), q = (int)(e - (long long)i * pieces);
    if (i >= live_rows(n_dev, cap)) return;
    out[(size_t)i * pieces + q] = src[(size_t)perm[i] * pieces + q];
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

// ---- gather-GEMM, round 3: pair-compacted tiles ----------------------------------------------------------------------
// The round-1/2 kernel above multiplies a block's 32 sites through EVERY tap that any of them has (13.3 of 27 taps are real on
// 64-line sweeps: half of the matrix-core work was zero rows) and walks gather -> LDS -> MFMA as one serial chain per wave.
// This formulation:
//   * a block owns M consecutive output sites.  For each tap the (input row, local output) pairs that EXIST are compacted
//     (wave ballot + prefix popcount) into an LDS list and cut into tiles of 16 pairs; only real pairs (plus < 16 pad rows per
//     tap) reach the matrix cores;
//   * a stage = TPS tiles.  All 256 threads gather the stage's input rows (16-B loads, one row = CIN*4 contiguous bytes) into
//     registers WHILE the previous stage is multiplied, then drop them into the other half of a double-buffered LDS tile: one
//     block barrier per stage, global latency off the critical path;
//   * work unit = (tile, 16 output channels).  Wave w takes units w, w+4, ...: with >= 64 output channels a wave owns its
//     channel slices exclusively (the tap's weight slice is read once per block, straight from the reference [K,Cin,Cout]
//     layout into B fragments, kept while the tap does not change); narrower layers give each group of waves its own copy of
//     the accumulator tile, summed in a fixed order at the end;
//   * a tile's 16 x 16 result rows belong to 16 DIFFERENT output sites: they are added (ds_add_f32, no return) to the block's
//     [M, COUT] accumulator in LDS.  No two waves ever add to the same word and a wave's LDS operations execute in order, so the
//     summation order is a fixed function of the input: bit-reproducible, no cross-wave atomics;
//   * A fragments with ONE ds_read_b128 per four k-steps: lane (row, g) reads channels 16j+4g..+3, i.e. k-step (j,i) multiplies
//     channel 16j+4g+i -- a permutation of the reduction order that the B fragment loads mirror; row stride CIN+8 words makes
//     the 16-lane groups of a b128 read conflict-free;
//   * BatchNorm scale/shift + ReLU and 16-B coalesced row stores in the epilogue.
template <int V> struct ILog2 { static constexpr int v = 1 + ILog2<V / 2>::v; };
template <> struct ILog2<1> { static constexpr int v = 0; };

template <int CIN, int COUT, int M, int TPS, int DB /*double-buffered gather tile*/, int DBG = 0>
__global__ __launch_bounds__(256) void k_sp_conv2(const float* __restrict__ feat_in, const int* __restrict__ nbr,
                                                  int out_cap, const int* __restrict__ n_dev, int K,
                                                  const float* __restrict__ wfrag /*heal_sp_weight_fragments*/,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  int relu, float* __restrict__ feat_out /*[n_out][COUT]*/) {
    static_assert(CIN % 4 == 0 && (CIN < 16 || CIN % 16 == 0) && COUT % 16 == 0 && M % 64 == 0, "shape");
    constexpr int LB = ILog2<M>::v;              // pair word = (input row << LB) | local output site
    constexpr int KC = CIN / 4;                  // k-steps
    constexpr int J = CIN >= 16 ? CIN / 16 : 0;  // b128 fragment reads per tile (wide rows)
    constexpr int JW = CIN >= 16 ? J : 1;        // 16-B weight fragment registers per set
    constexpr int NC = COUT / 16;                // 16-channel output slices
    constexpr int COPIES = NC >= 4 ? 1 : 4 / NC; // accumulator copies (narrow layers: one per group of waves)
    constexpr int UNITS = TPS * NC;
    static_assert(UNITS % 4 == 0 && (NC >= 4 ? NC % 4 == 0 : 4 % NC == 0), "unit split");
    constexpr int UPW = UNITS / 4;               // units per wave and stage
    constexpr int ROWS = TPS * 16;               // gathered rows per stage
    constexpr int CPR = CIN / 4;                 // 16-B chunks per row
    constexpr int CHUNKS = ROWS * CPR;
    static_assert(CHUNKS % 256 == 0, "gather split");
    constexpr int CPT = CHUNKS / 256;            // DMA instructions per wave and stage
    constexpr int CH = M / 64;                   // 64-site chunks of the block
    constexpr int TPT = M / 16;                  // tiles per tap, worst case
    constexpr int MAXT = 27 * TPT;               // tiles, worst case
    constexpr uint32_t PAD = 0xFFFFFFFFu;
    constexpr int RSA = COUT + 4;                // accumulator row stride (words): 16-B slots of different rows spread over the banks

    __shared__ uint32_t s_pair[27 * M + 16 * 3 * TPS];
    __shared__ __attribute__((aligned(16))) float s_acc[COPIES * M * RSA];
    __shared__ __attribute__((aligned(1024))) float s_A[(DB ? 2 : 1) * ROWS * CIN];
    __shared__ int s_tap[MAXT + 3 * TPS];        // tap of every tile
    __shared__ int s_cnt[32];

    const int n_out = live_rows(n_dev, out_cap);
    // XCD-contiguous site ranges: neighbouring blocks share gathered rows, keep them in one L2.  The permutation is over the
    // LIVE blocks: with device-side row counts the grid is capacity-sized, and a permutation of the whole grid would hand every
    // live block to the first XCDs (measured: 2.4x slower inside the captured pipeline than stand-alone)
    const unsigned nb_ = (unsigned)(n_out + M - 1) / M;
    if (blockIdx.x >= nb_) return;
    const unsigned q_ = nb_ >> 3, r_ = nb_ & 7u, x_ = blockIdx.x & 7u;
    const int blk = (int)((x_ < r_ ? x_ * (q_ + 1) : r_ * (q_ + 1) + (x_ - r_) * q_) + (blockIdx.x >> 3));
    const int site0 = blk * M;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int g = l >> 4, ln = l & 15;
    const int swz = (ln * CPR) >> 4;             // XOR swizzle of the 16-B chunks of gathered row ln (see the gather)

    // ---- zero the accumulator tile(s) ------------------------------------------------------------------------------
    for (int i = tid; i < COPIES * M * RSA / 4; i += 256)
        reinterpret_cast<float4*>(s_acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- compact the (input row, local site) pairs of every tap, tap after tap, each tap padded to whole tiles -------
    // pass 1: neighbour ids -> registers, pair counts per tap; pass 2 (after the prefix sum over the taps): the lists.
    // Tile t is then simply s_pair[16 t .. 16 t + 15], a stage the TPS tiles behind tile stage * TPS (taps may change inside a
    // stage: strided layers have a handful of pairs per tap).
    int T;
    {
        int v[7][CH];
#pragma unroll
        for (int ti = 0; ti < 7; ++ti) {
            const int t = wave + 4 * ti;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int site = site0 + c * 64 + l;
                v[ti][c] = (t < K && site < n_out) ? nbr[(size_t)site * K + t] : -1;
            }
        }
#pragma unroll
        for (int ti = 0; ti < 7; ++ti) {
            const int t = wave + 4 * ti;
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) cnt += __popcll(__ballot(v[ti][c] >= 0));
            if (t < K && l == 0) s_cnt[t] = cnt;
        }
        __syncthreads();
        const int mine = l < K ? (s_cnt[l] + 15) >> 4 : 0;       // tiles of tap l
        const int incl = wave_incl_scan(mine);
        T = __shfl(incl, 63, 64);
#pragma unroll
        for (int ti = 0; ti < 7; ++ti) {
            const int t = wave + 4 * ti;
            if (t < K) {
                const int nt = __shfl(mine, t, 64), t0 = __shfl(incl, t, 64) - nt;   // first tile of tap t
                int base = 16 * t0;
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const unsigned long long have = __ballot(v[ti][c] >= 0);
                    if (v[ti][c] >= 0)
                        s_pair[base + __popcll(have & lanemask_lt())] = ((uint32_t)v[ti][c] << LB) | (uint32_t)(c * 64 + l);
                    base += __popcll(have);
                }
                if (l < 16 && base + l < 16 * (t0 + nt)) s_pair[base + l] = PAD;
                if (l < nt) s_tap[t0 + l] = t;
            }
        }
        // the missing tiles of the last stage + the two stages past the end that the prefetches touch: all-pad tiles of the
        // last tap (no weight reload)
        if (wave == 3) {
            const int last_tap = 63 - __builtin_clzll(__ballot(mine > 0) | 1ull);
            for (int i = l; i < 16 * 3 * TPS; i += 64) s_pair[16 * T + i] = PAD;
            if (l < 3 * TPS) s_tap[T + l] = last_tap;
        }
    }
    __syncthreads();
    const int n_stages = (T + TPS - 1) / TPS;

    // ---- the gather: global -> LDS by DMA -----------------------------------------------------------------------------------
    // global_load_lds_dwordx4: no staging registers, no ds_write pass, in flight during the MFMAs.  Instruction k = wave + 4 i
    // of a stage writes the 1 KiB behind tile byte 1024 k, lane L its 16 B number L -- row (64 k + L) / CPR, slot L % CPR.  The
    // LDS image is lane-linear, so the bank-conflict fix is an XOR swizzle applied on the SOURCE side (the lane fetches chunk
    // slot ^ f(row)) and again on the fragment reads.  Pad rows and the stages past the end read row 0; their products are
    // never accumulated.  The source rows of a stage are looked up one stage before its loads are issued.
    // The DMA and the weight loads are issued from asm statements on purpose: hipcc's waitcnt pass treats a counted
    // global_load_lds as a possible writer of every LDS word and as a reason to drain vmcnt(0) at the next use of ANY ordinary
    // load, and a counted load behind a branch made it wait in the middle of the MFMA sequence.  Hidden from it, they cost no
    // wait until the explicit vmcnt(0) in front of the stage's closing barrier.  The DMA has no VGPR destination; the weight
    // registers are only read behind HEAL_SP_W_WAIT, which takes them "+v".  M0 is saved and restored inside the statement.
    int g_row[CPT];
    uint32_t g_src[CPT], g_chunk[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        g_row[i] = (64 * (wave + 4 * i) + l) / CPR;
        g_chunk[i] = (uint32_t)((l % CPR) ^ (((g_row[i] & 15) * CPR) >> 4));
    }
    const uint32_t sA_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)s_A;
#define HEAL_SP_GATHER_INDEX(stage)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                                                  \
        const uint32_t e_ = s_pair[(stage) * ROWS + g_row[i]];                                                         \
        g_src[i] = e_ == PAD ? 0u : e_ >> LB;                                                                          \
    }                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < UPW; ++q)                                                                    \
        tapn[q] = __builtin_amdgcn_readfirstlane(s_tap[(stage) * TPS + (wave + 4 * q) / NC]);
#define HEAL_SP_GATHER_ISSUE(buf_)                                                                                     \
    if constexpr (!(DBG & 2)) {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                                                              \
            const float* gsrc_ = feat_in + (size_t)g_src[i] * CIN + g_chunk[i] * 4;                                    \
            const uint32_t dst_ = __builtin_amdgcn_readfirstlane(sA_lds + (uint32_t)(((buf_) * ROWS * CIN + (wave + 4 * i) * 256) * 4)); \
            uint32_t keep_;                                                                                            \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                         : "=&s"(keep_) : "v"(gsrc_), "s"(dst_) : "memory");                                           \
        }                                                                                                              \
    }
// (counted forms -- the weight loads left in flight across the barrier, vmcnt(CPT) in front of the MFMAs -- measured slower than
// one drain at the end of the stage: both prefetches have had the whole stage to land by then)
#define HEAL_SP_VMCNT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
    // weight fragments of unit c for tap `tap_` (channel slice nb = (wave + 4 c) % NC): lane (cout ln, k-group g) holds
    // W[tap][kperm(step, g)][16 nb + ln], pre-laid by heal_sp_weight_fragments as [tap][nb][j][lane][4] (16-B loads, 1 KiB
    // contiguous per instruction)
#define HEAL_SP_LOAD_W(dst_, c, tap_)                                                                                  \
    {                                                                                                                  \
        const float* wt_ = wfrag + ((size_t)(tap_) * NC + (wave + 4 * (c)) % NC) * (CIN * 16) + l * (CIN >= 16 ? 4 : KC); \
        if constexpr (CIN >= 16) {                                                                                     \
            _Pragma("unroll") for (int j = 0; j < J; ++j)                                                              \
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_[c][j]) : "v"(wt_ + j * 256) : "memory");    \
        } else {                                                                                                       \
            _Pragma("unroll") for (int kc = 0; kc < KC; ++kc)                                                          \
                asm volatile("global_load_dword %0, %1, off" : "=v"(dst_##s[c][kc]) : "v"(wt_ + kc) : "memory");       \
        }                                                                                                              \
    }
    // (every asm destination is a WHOLE register variable: an element of a vector would make the compiler copy the scalar asm
    // output into the vector right behind the load, i.e. read the register before the data has landed)
#define HEAL_SP_W_WAIT(w_, c, n_)                                                                                     \
    if constexpr (CIN >= 16) {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < JW; ++j)                                                                \
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w_[c][j]) : "n"(n_) : "memory");                                \
    } else {                                                                                                          \
        _Pragma("unroll") for (int kc = 0; kc < KC; ++kc)                                                             \
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w_##s[c][kc]) : "n"(n_) : "memory");                            \
    }
    constexpr int NWL = (DBG & 8) ? 0 : UPW * (CIN >= 16 ? J : KC);   // weight load instructions per stage

    f32x4 wfr[UPW][JW], wnx[UPW][JW];   // wide layers: 16-B fragment registers, one set per unit (wnx: the asm destinations)
    float wfrs[UPW][KC], wnxs[UPW][KC]; // CIN < 16: KC <= 2 scalars
    int tapn[UPW];                      // taps of the next stage's tiles
    const int copy = NC >= 4 ? 0 : wave / NC;
    float* acc_base = s_acc + (size_t)copy * M * RSA;

    // Every asm load and its wait statement sit inside ONE loop iteration: a loop-carried asm destination would get a compiler
    // copy on the back edge, i.e. a read of the register before its data has landed.
    f32x4 acc[UPW];
    float xfr[UPW][KC];
    uint32_t lw[UPW];
    // fragments of the gathered rows of `stage_` (LDS buffer buf_), the pair words of this wave's tiles
#define HEAL_SP_READ_STAGE(stage_, buf_)                                                                               \
    _Pragma("unroll") for (int q = 0; q < UPW; ++q) {                                                                  \
        const int ti = (wave + 4 * q) / NC;                                                                            \
        const float* a = &s_A[((buf_) * ROWS + ti * 16 + ln) * CIN];                                                   \
        if constexpr (CIN >= 16) {                                                                                     \
            _Pragma("unroll") for (int j = 0; j < J; ++j) {                                                            \
                const float4 t4 = *reinterpret_cast<const float4*>(a + (((4 * j + g) ^ swz) << 2));                    \
                xfr[q][4 * j] = t4.x; xfr[q][4 * j + 1] = t4.y; xfr[q][4 * j + 2] = t4.z; xfr[q][4 * j + 3] = t4.w;    \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int kc = 0; kc < KC; ++kc) xfr[q][kc] = a[((kc ^ swz) << 2) + g];                   \
        }                                                                                                              \
        lw[q] = s_pair[((stage_) * TPS + ti) * 16 + ln];                                                               \
    }
    // MFMAs (the units of a wave are independent accumulator chains).  D^T[cout][pair] = W^T X^T: weights are the A operand,
    // gathered rows the B operand, so a lane ends up with FOUR consecutive output channels of ONE pair = one 16-B
    // read-modify-write of the accumulator row.  Units beyond the last stage's tiles multiply pad rows; nothing of them is kept.
#define HEAL_SP_MFMA_STAGE()                                                                                           \
    _Pragma("unroll") for (int q = 0; q < UPW; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};                              \
    _Pragma("unroll") for (int kc = 0; kc < KC; ++kc)                                                                  \
        _Pragma("unroll") for (int q = 0; q < UPW; ++q) {                                                              \
            const float wv = CIN >= 16 ? wfr[q][CIN >= 16 ? kc / 4 : 0][kc % 4] : wfrs[q][kc % KC];                    \
            if constexpr (DBG & 4) acc[q][0] += xfr[q][kc] * wv;                                                       \
            else acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xfr[q][kc], acc[q], 0, 0, 0);                       \
        }
    // the gathered rows of the stage whose source rows g_src holds; single buffer: every wave must hold its fragments first
#define HEAL_SP_NEXT_GATHER(buf_)                                                                                      \
    if constexpr (DB) {                                                                                                \
        HEAL_SP_GATHER_ISSUE(buf_)                                                                                     \
    } else {                                                                                                           \
        lds_barrier();                                                                                                 \
        HEAL_SP_GATHER_ISSUE(0)                                                                                        \
    }
    // the prefetched weights have landed -> the MFMA operand registers
#define HEAL_SP_TAKE_W()                                                                                               \
    if constexpr (NWL) {                                                                                               \
        _Pragma("unroll") for (int q = 0; q < UPW; ++q) {                                                              \
            HEAL_SP_W_WAIT(wnx, q, 0)                                                                                  \
            _Pragma("unroll") for (int j = 0; j < JW; ++j) wfr[q][j] = wnx[q][j];                                      \
            _Pragma("unroll") for (int kc = 0; kc < KC; ++kc) wfrs[q][kc] = wnxs[q][kc];                               \
        }                                                                                                              \
    }

    HEAL_SP_GATHER_INDEX(0)
    HEAL_SP_GATHER_ISSUE(0)
    if constexpr (NWL) {
#pragma unroll
        for (int q = 0; q < UPW; ++q) HEAL_SP_LOAD_W(wnx, q, tapn[q])
    }
    HEAL_SP_GATHER_INDEX(1)
    HEAL_SP_VMCNT(0);
    HEAL_SP_TAKE_W()
    lds_barrier();
    for (int s = 0; s < ((DBG & 16) ? 0 : n_stages); ++s) {
        const int buf = DB ? (s & 1) : 0;
        HEAL_SP_READ_STAGE(s, buf)
        // ---- prefetch for stage s+1: the weight fragments of its taps, then the gathered rows.  The weight loads are
        // unconditional: a tap that stays is an L1 hit, and every conditional form measured slower than the reload
        if constexpr (NWL) {
#pragma unroll
            for (int q = 0; q < UPW; ++q) HEAL_SP_LOAD_W(wnx, q, tapn[q])
        }
        HEAL_SP_NEXT_GATHER(buf ^ 1)
        HEAL_SP_MFMA_STAGE()
        HEAL_SP_GATHER_INDEX(s + 2)
        // ---- add the tiles' rows to the block accumulator (exclusive owner: plain read-modify-write, in program order) ------------
#pragma unroll
        for (int q = 0; q < UPW; ++q) {
            const int nb = (wave + 4 * q) % NC;
            if (lw[q] != PAD && !(DBG & 1)) {
                float4* p = reinterpret_cast<float4*>(&acc_base[(lw[q] & (uint32_t)(M - 1)) * RSA + nb * 16 + 4 * g]);
                float4 o = *p;
                o.x += acc[q][0]; o.y += acc[q][1]; o.z += acc[q][2]; o.w += acc[q][3];
                *p = o;
            }
            if ((DBG & 1) && acc[q][0] == 12345.678f) acc_base[0] = 1.f;
        }
        HEAL_SP_VMCNT(0);   // the DMA of stage s+1 and its weights
        HEAL_SP_TAKE_W()
        lds_barrier();
    }
#undef HEAL_SP_READ_STAGE
#undef HEAL_SP_MFMA_STAGE
#undef HEAL_SP_NEXT_GATHER
#undef HEAL_SP_TAKE_W
    // ---- epilogue: sum the copies in a fixed order, BatchNorm1d (eval) + ReLU, 16-B row stores -----------------------------
    constexpr int C4 = COUT / 4;
    for (int i = tid; i < M * C4; i += 256) {
        const int m = i / C4, c4 = i - m * C4;
        const int site = site0 + m;
        if (site >= n_out) break;
        float4 v = *reinterpret_cast<const float4*>(&s_acc[m * RSA + c4 * 4]);
#pragma unroll
        for (int k = 1; k < COPIES; ++k) {
            const float4 t4 = *reinterpret_cast<const float4*>(&s_acc[(k * M + m) * RSA + c4 * 4]);
            v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
        }
        const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(feat_out + (size_t)site * COUT + c4 * 4) = v;
    }
#undef HEAL_SP_GATHER_INDEX
#undef HEAL_SP_GATHER_ISSUE
#undef HEAL_SP_VMCNT
#undef HEAL_SP_LOAD_W
#undef HEAL_SP_W_WAIT
}

// ---- gather-GEMM, round 6: the thin layers (CIN <= 16) on the pair-tile rulebook ------------------------------------------------
// On the 4 -> 16, 16 -> 16 and 16 -> 32 layers of VoxelBackBone8x (sparse_backbone_3d.py:48-62) a tile of 16 pairs is ONE to EIGHT
// matrix instructions: k_sp_conv2's block-wide stages (compaction of the [site][tap] table, barrier + drain per 128-256 gathered
// rows, 4 blocks per CU) leave such a layer waiting -- 0.14 of the HBM roof.  Here the rulebook already IS the list of pair tiles
// (k_sp_nbr_tiles below: one fixed-stride slot per S output sites, only its used prefix written and read), ONE WAVE owns a slot
// and never meets a barrier, and three pipeline stages run straight from global memory: while group i (2 tiles) multiplies, the
// input rows and weight fragments of group i+1 and the pair words of group i+2 are in flight.
//   * pair word = input row << SPT_LB(S) | accumulator row; padding pairs read input row 0 and add into a trash row (S) of the
//     accumulator: no predicate, no branch anywhere in the loop; the slot's tile count is a multiple of 4 (all-padding tiles);
//   * lane (pair, g) reads channels 4g..4g+3 of its pair's input row (one 16-B load; CIN = 4: one dword, channel g): the B
//     fragment as it lies; the tap's weight fragment comes from heal_sp_weight_fragments' layout (L1 / L2 hits); KC x NC MFMAs;
//   * a tile's 16 x 16 result rows belong to 16 different output sites: one 16-B read-modify-write per lane and slice of the
//     wave's private [S + 1][COUT] accumulator in LDS.  LDS operations of one wave execute in order and nothing is shared
//     between waves: a fixed summation order (taps ascending per site), bit-reproducible, no atomics;
//   * LDS holds the accumulator only (5-18 KB; the tap bytes travel with the pair words), the register file sets the occupancy.
// S = 64 or 128 output sites per slot (the caller's choice; measured, profiles/r06_k3_thin.json): 128 sites fill the 16-pair tiles
// better (0.73 -> 0.84 on the submanifold layers, 0.5 -> 0.68 on the first strided layer with its 2.5 live taps of 27 per site) and
// save 5 us on that layer, but k_sp_nbr_tiles<128> has half the waves for the same lookups and loses 22 us: 64 is the default.
// What bounds the kernel (PMC, profiles/r06_k3_tiles_pmc.txt): a 20-50 us kernel of 5 300 one-wave blocks whose waves live 12-15 us
// each -- launch ramp, a chain of three dependent round trips before the first MFMA, and the matrix pipe itself (24 M busy cycles =
// 10 us per SIMD on 16 -> 16: a 16-pair fp32 tile is 4 x 32 cycles whether 6 or 16 of its rows are real).
constexpr int SPT_HDR_WORDS = 64;                            // word 0: tiles T (multiple of 4); bytes 4 .. 4 + T + 4: tap of tile i
//   (behind the T tiles one more group of 4 all-padding tiles, not counted: what the consumer's pipeline reads past the end)
__host__ __device__ constexpr int spt_lb(int S) { return S == 64 ? 7 : 8; }
__host__ __device__ constexpr int spt_slot_words(int S) { return SPT_HDR_WORDS + 27 * S + 128; }   // + padding to 4 tiles + one all-padding group

template <int CIN, int COUT, int S, int DBG = 0, int D = 2>
__global__ __launch_bounds__(64) void k_sp_tiles(const float* __restrict__ feat_in, const uint32_t* __restrict__ tiles, int out_cap,
                                                 const int* __restrict__ n_dev, const float* __restrict__ wfrag,
                                                 const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                 float* __restrict__ feat_out) {
    static_assert((CIN == 4 || CIN == 16 || CIN == 32) && COUT % 16 == 0 && (S == 64 || S == 128), "shape");
    constexpr int J = CIN >= 16 ? CIN / 16 : 1;   // 16-B pieces of an input row per lane (= of a weight fragment per slice)
    static_assert(D == 4 || D == 2, "tiles per pipeline group");
    constexpr int LB = spt_lb(S);
    constexpr int NC = COUT / 16;
    constexpr int RSA = COUT + 4;                // accumulator row stride (words)
    constexpr int MAXT = 27 * S / 16 + 4;
    constexpr int SLOT = spt_slot_words(S);
    __shared__ __attribute__((aligned(16))) float s_acc[(S + 1) * RSA];

    const int n_out = live_rows(n_dev, out_cap);
    const unsigned nb_ = (unsigned)(n_out + S - 1) / S;
    if (blockIdx.x >= nb_) return;
    // XCD-contiguous site ranges over the LIVE slots (see k_sp_conv2)
    const unsigned q_ = nb_ >> 3, r_ = nb_ & 7u, x_ = blockIdx.x & 7u;
    const int blk = (int)((x_ < r_ ? x_ * (q_ + 1) : r_ * (q_ + 1) + (x_ - r_) * q_) + (blockIdx.x >> 3));
    const int site0 = blk * S;
    const int l = threadIdx.x, g = l >> 4, ln = l & 15;
    const uint32_t* slot = tiles + (size_t)blk * SLOT;
    const uint32_t* list = slot + SPT_HDR_WORDS + ln;
    // the first group's pair words and tap bytes do not depend on the tile count (an empty slot still holds its all-padding group):
    // they are requested TOGETHER with it -- two dependent round trips in front of the first MFMA instead of three
    uint32_t e0[D];
#pragma unroll
    for (int q = 0; q < D; ++q) e0[q] = list[q * 16];
    uint32_t tw0 = slot[1];
    const int T = min((int)slot[0], MAXT) & ~3;
    for (int i = l; i < (S + 1) * RSA / 4; i += 64) reinterpret_cast<float4*>(s_acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const char* fin = reinterpret_cast<const char*>(feat_in) + (CIN >= 16 ? 16 * g : 4 * g);
    const char* wf = reinterpret_cast<const char*>(wfrag) + (CIN >= 16 ? 16 * l : 4 * l);
    constexpr uint32_t WTAP = NC * (CIN >= 16 ? J * 1024u : 256u);   // bytes of one tap's fragments
    struct Xw {
        f32x4 x[D][J];              // CIN >= 16: channels 16j+4g..+3 of the pair's input row; CIN = 4: .x = channel g
        f32x4 w[D][NC][J];          // CIN >= 16: the four k-steps (16j+4g+i) of (tap, slice); CIN = 4: .x
    };
    // stage 1: pair words + the header word with the group's tap bytes (4 per word; past the end: the all-padding group)
    auto load_e = [&](uint32_t (&e)[D], uint32_t& tw, int base) {
        const int b_ = min(base, T);
        const uint32_t* p = list + b_ * 16;
#pragma unroll
        for (int q = 0; q < D; ++q) e[q] = p[q * 16];
        tw = slot[1 + (b_ >> 2)];
    };
    auto load_xw = [&](const uint32_t (&e)[D], uint32_t tw, Xw& G, int base) {   // stage 2: input rows + weight fragments
        const int b_ = min(base, T);
        const uint32_t taps = (DBG & 2) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)tw) >> (8 * (b_ & 3));
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const uint32_t row = (DBG & 1) ? 0u : e[q] >> LB;
            const char* wt = wf + ((taps >> (8 * q)) & 0xFFu) * WTAP;
            if constexpr (CIN >= 16) {
#pragma unroll
                for (int j = 0; j < J; ++j) G.x[q][j] = *reinterpret_cast<const f32x4*>(fin + row * (4u * CIN) + 64 * j);
#pragma unroll
                for (int nb = 0; nb < NC; ++nb)
#pragma unroll
                    for (int j = 0; j < J; ++j) G.w[q][nb][j] = *reinterpret_cast<const f32x4*>(wt + (nb * J + j) * 1024);
            } else {
                G.x[q][0][0] = *reinterpret_cast<const float*>(fin + row * 16u);
#pragma unroll
                for (int nb = 0; nb < NC; ++nb) G.w[q][nb][0][0] = *reinterpret_cast<const float*>(wt + nb * 256);
            }
        }
    };
    auto multiply = [&](const uint32_t (&e)[D], const Xw& G) {   // stage 3
#pragma unroll
        for (int q = 0; q < D; ++q) {
            float* p = &s_acc[(e[q] & (uint32_t)((1 << LB) - 1)) * RSA + 4 * g];
#pragma unroll
            for (int nb = 0; nb < NC; ++nb) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int i = 0; i < (CIN >= 16 ? 4 : 1); ++i)
                        a = __builtin_amdgcn_mfma_f32_16x16x4f32(G.w[q][nb][j][i], G.x[q][j][i], a, 0, 0, 0);
                if constexpr (!(DBG & 4)) {
                    float4* p4 = reinterpret_cast<float4*>(p + nb * 16);   // (per-dword ds_add_f32 measured 8x slower: 8-way bank conflicts)
                    float4 o = *p4;
                    o.x += a[0]; o.y += a[1]; o.z += a[2]; o.w += a[3];
                    *p4 = o;
                } else if (a[0] == 12345.678f) p[0] = 1.f;
            }
        }
    };
    if (T > 0 && !(DBG & 8)) {
        // two register sets for the rows / weights, used alternately by the two halves of the loop body: a copy "C = B" would make
        // the compiler wait for the loads it has just issued
        uint32_t eA[D], eB[D], eC[D], twA, twB;
        Xw X0, X1;
#pragma unroll
        for (int q = 0; q < D; ++q) eB[q] = e0[q];
        twB = tw0;
        load_e(eA, twA, D);
        load_xw(eB, twB, X0, 0);
        for (int base = 0; base < T; base += 2 * D) {
#pragma unroll
            for (int q = 0; q < D; ++q) { eC[q] = eB[q]; eB[q] = eA[q]; }
            twB = twA;
            load_e(eA, twA, base + 2 * D);
            load_xw(eB, twB, X1, base + D);
            __builtin_amdgcn_sched_barrier(0);   // the loads are issued BEFORE the group that hides them multiplies
            multiply(eC, X0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < D; ++q) { eC[q] = eB[q]; eB[q] = eA[q]; }
            twB = twA;
            load_e(eA, twA, base + 3 * D);
            load_xw(eB, twB, X0, base + 2 * D);
            __builtin_amdgcn_sched_barrier(0);
            multiply(eC, X1);   // T / 4 odd: the all-padding group (input row 0 into the trash row)
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: BatchNorm1d (eval) + ReLU, 16-B row stores (the wave's rows are contiguous) -----------------------------------
    constexpr int C4 = COUT / 4;
#pragma unroll 2
    for (int i0 = 0; i0 < S * C4; i0 += 64) {
        const int i = i0 + l, m = i / C4, c4 = i - m * C4;
        if (site0 + m < n_out) {
            float4 o = *reinterpret_cast<const float4*>(&s_acc[m * RSA + c4 * 4]);
            const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
            const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
            o.x = fmaf(o.x, sc.x, sh.x); o.y = fmaf(o.y, sc.y, sh.y); o.z = fmaf(o.z, sc.z, sh.z); o.w = fmaf(o.w, sc.w, sh.w);
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(feat_out + (size_t)(site0 + m) * COUT + c4 * 4) = o;
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

// ---- rank structure of a site set: occupancy bitmap + prefix counts --------------------------------------------------------
// For the site sets the encoder PRODUCES (the outputs of its strided convolutions) the hash grid, the candidate dedup, the radix
// sort and the hash probes of the neighbour search are all replaced by one dense structure over the output grid
// (cells / 8 bytes: 22 MB at [8 x 21 x 1024 x 1024], nothing next to 288 GB):
//   * k_sp_bm_mark    every input site sets the bits of the <= 8 output cells whose receptive field contains it;
//   * k_sp_bm_count   population count per granule of 256 cells, exclusive scan -> base[granule], total = site count;
//   * k_sp_bm_emit    walks the bitmap in order: the sites come out SORTED by linear coordinate -- no sort;
//   * rank(key) = base[key / 256] + popcount of the granule's bits below key = the site's row.  The three x taps of a kernel row
//     fall into one 32-B granule, and neighbouring sites query neighbouring granules (the hash scattered every probe).
// The root site set (the voxels, a 1.4 G-cell grid at 0.1 m) keeps the hash grid.
constexpr int SP_GRAN = 256;   // cells per granule (8 words)

struct SpRank {
    const uint32_t* bm;
    const int* base;
    const uint32_t* l1;   // root structure only: one bit per granule; granules whose bit is clear hold stale words
};

__device__ __forceinline__ int sp_rank_lookup(const SpRank& r, uint32_t key) {
    const uint32_t w = key >> 5;
    if (r.l1 && !((r.l1[key >> 13] >> ((key >> 8) & 31u)) & 1u)) return -1;
    const uint32_t word = r.bm[w];
    const uint32_t bit = key & 31u;
    if (!((word >> bit) & 1u)) return -1;
    const uint32_t g = key >> 8;
    const uint4* gp = reinterpret_cast<const uint4*>(r.bm + ((size_t)g << 3));
    const uint4 a = gp[0], b = gp[1];
    const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const int wi = (int)(w & 7u);
    int c = __popc(word & ((1u << bit) - 1u));
#pragma unroll
    for (int i = 0; i < 7; ++i) c += i < wi ? __popc(ws[i]) : 0;
    return r.base[g] + c;
}

__global__ __launch_bounds__(256) void k_sp_bm_mark(const int4* __restrict__ in_idx, int in_cap,
                                                   const int* __restrict__ n_dev, SpConvGeom g,
                                                   uint32_t* __restrict__ bm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, in_cap)) return;
    const int4 c = in_idx[i];
    // output o sees input i through tap k = i + p - o s, 0 <= k < K  ->  o in [ceil((i + p - K + 1) / s), floor((i + p) / s)]
    int lo[3], hi[3];
    const int ci[3] = {c.y, c.z, c.w};
    const int od[3] = {g.out.D, g.out.H, g.out.W};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a = ci[d] + g.p[d];
        const int num = a - g.k[d] + 1;
        lo[d] = num <= 0 ? 0 : (num + g.s[d] - 1) / g.s[d];
        hi[d] = min(a / g.s[d], od[d] - 1);
    }
    for (int oz = lo[0]; oz <= hi[0]; ++oz)
        for (int oy = lo[1]; oy <= hi[1]; ++oy)
            for (int ox = lo[2]; ox <= hi[2]; ++ox) {
                const uint32_t key = sp_key(g.out, c.x, oz, oy, ox);
                const uint32_t bit = 1u << (key & 31u);
                if (!(bm[key >> 5] & bit)) atomicOr(&bm[key >> 5], bit);
            }
}

__global__ __launch_bounds__(256) void k_sp_bm_count(const uint4* __restrict__ bm4, int granules, int* __restrict__ cnt) {
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= granules) return;
    const uint4 a = bm4[2 * (size_t)gi], b = bm4[2 * (size_t)gi + 1];
    cnt[gi] = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
}

// one lane per bitmap word, eight lanes per granule: the sites leave in ascending key order
__global__ __launch_bounds__(256) void k_sp_bm_emit(const uint32_t* __restrict__ bm, const int* __restrict__ base,
                                                   size_t words, int out_cap, SpShape s, int4* __restrict__ out_idx,
                                                   const int* __restrict__ n_out, int* __restrict__ overflow) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    // sticky capacity report: a captured graph re-creates n_out on every replay, this word only ever grows
    if (overflow && w == 0 && *n_out > out_cap) atomicMax(overflow, *n_out);
    uint32_t word = w < words ? bm[w] : 0u;
    const int mine = __popc(word);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const int t = __shfl_up(incl, o, 8);
        if ((threadIdx.x & 7) >= o) incl += t;
    }
    if (!word) return;
    int pos = base[w >> 3] + incl - mine;
    while (word) {
        const int b = __ffs(word) - 1;
        word &= word - 1u;
        if (pos < out_cap) {
            uint32_t k = (uint32_t)(w << 5) + (uint32_t)b;
            const int x = k % s.W; k /= s.W;
            const int y = k % s.H; k /= s.H;
            const int z = k % s.D; k /= s.D;
            out_idx[pos] = make_int4((int)k, z, y, x);
        }
        ++pos;
    }
}

// neighbour rows through the rank structure of the INPUT site set: one thread per (output site, kz, ky), the kx taps of a
// kernel row are consecutive keys
__global__ __launch_bounds__(256) void k_sp_nbr_rank(const int4* __restrict__ out_idx, int out_cap,
                                                    const int* __restrict__ n_dev, SpConvGeom g, SpRank r, int in_cap,
                                                    const int* __restrict__ n_in_dev, int* __restrict__ nbr) {
    const int KR = g.k[0] * g.k[1], KX = g.k[2];
    const int n_in = live_rows(n_in_dev, in_cap);   // sites beyond the capacity of the input set were never emitted
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)live_rows(n_dev, out_cap) * KR) return;
    const int o = (int)(t / KR), kr = (int)(t - (long long)o * KR);
    const int kz = kr / g.k[1], ky = kr - kz * g.k[1];
    const int4 c = out_idx[o];
    const int z = c.y * g.s[0] - g.p[0] + kz;
    const int y = c.z * g.s[1] - g.p[1] + ky;
    const int x0 = c.w * g.s[2] - g.p[2];
    int* dst = nbr + (size_t)o * (KR * KX) + kr * KX;
    const bool line_ok = z >= 0 && z < g.in.D && y >= 0 && y < g.in.H;
    const uint32_t line = line_ok ? sp_key(g.in, c.x, z, y, 0) : 0u;
    for (int kx = 0; kx < KX; ++kx) {
        const int x = x0 + kx;
        int row = (line_ok && x >= 0 && x < g.in.W) ? sp_rank_lookup(r, line + (uint32_t)x) : -1;
        dst[kx] = row < n_in ? row : -1;
    }
}

// ---- rank structure of the ROOT site set (the voxels) ---------------------------------------------------------------------------------
// The voxel grid of SECOND is 1.4e9 cells at 0.1 m (172 MB of bitmap) with 3e5 active sites: clearing or scanning the bitmap costs
// more than everything else, so the root set gets a TWO-LEVEL structure: `l1` has one bit per 256-cell granule (0.7 MB, the only
// thing cleared per frame); a granule of the big bitmap is valid only if its l1 bit is set -- the thread that sets the bit zeroes
// the granule, a second launch then ORs the cell bits in.  Counting, the prefix sums and the per-granule bases walk l1, i.e. touch
// only the occupied granules.  rank(key) is the site's row in linear-coordinate order: the sites are "sorted" by one scatter (no
// radix sort: 4 x (histogram + scatter) = 200 us for 3.4e5 sites), and the neighbour queries of the first two layers use
// k_sp_nbr_rank instead of 9e6 hash probes (135 us each).
__global__ __launch_bounds__(256) void k_sp_root_mark1(const int4* __restrict__ idx, int cap, const int* __restrict__ n_dev,
                                                      SpShape s, uint32_t* __restrict__ l1, uint32_t* __restrict__ bm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const int4 c = idx[i];
    const uint32_t g = sp_key(s, c.x, c.y, c.z, c.w) >> 8;
    // EVERY site wipes its granule (the cell bits are ORed in by the next launch, so within this one all writers write zeros) and
    // sets the directory bit with a fire-and-forget atomic: no returning atomic, no "who was first" (39 -> see DESIGN 3)
    uint4* gp = reinterpret_cast<uint4*>(bm + ((size_t)g << 3));
    gp[0] = make_uint4(0u, 0u, 0u, 0u);
    gp[1] = make_uint4(0u, 0u, 0u, 0u);
    atomicOr(&l1[g >> 5], 1u << (g & 31u));
}

__global__ __launch_bounds__(256) void k_sp_root_mark2(const int4* __restrict__ idx, int cap, const int* __restrict__ n_dev,
                                                      SpShape s, uint32_t* __restrict__ bm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const int4 c = idx[i];
    const uint32_t key = sp_key(s, c.x, c.y, c.z, c.w);
    atomicOr(&bm[key >> 5], 1u << (key & 31u));
}

__device__ __forceinline__ int sp_gran_count(const uint32_t* __restrict__ bm, uint32_t g) {
    const uint4* gp = reinterpret_cast<const uint4*>(bm + ((size_t)g << 3));
    const uint4 a = gp[0], b = gp[1];
    return __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
}

// One thread per directory word (32 granules; most words are empty: the loop runs over the set bits), 256 words per block.
// Pass 0: sites below every word -> cnt[w], block totals -> blk_sum, totals of groups of SP_ROOT_GROUP blocks -> grp_sum.
// Pass 1: base[g] = sites in front of granule g for the occupied granules: whole groups + the blocks of the own group + the words
// in front inside the block (a block scan) + the granules in front inside the word -- no separate scan launches.
constexpr int SP_ROOT_GROUP = 64;   // blocks per coarse sum
template <int PASS>
__global__ __launch_bounds__(256) void k_sp_root_count(const uint32_t* __restrict__ l1, const uint32_t* __restrict__ bm, int l1_words,
                                                      int* __restrict__ cnt, int* __restrict__ blk_sum, int* __restrict__ grp_sum,
                                                      int* __restrict__ base) {
    __shared__ int s_w[4], s_red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * 256 + threadIdx.x;
    uint32_t m = w < l1_words ? l1[w] : 0u;
    if constexpr (PASS == 0) {
        int c = 0;
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1u;
            c += sp_gran_count(bm, (uint32_t)w * 32u + (uint32_t)j);
        }
        if (w < l1_words) cnt[w] = c;
        c = wave_sum_i(c);
        if (lane == 0) s_w[wave] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            blk_sum[blockIdx.x] = t;
            if (t) atomicAdd(&grp_sum[blockIdx.x / SP_ROOT_GROUP], t);     // cleared together with the directory
        }
    } else {
        const int c = w < l1_words ? cnt[w] : 0;
        const int incl = wave_incl_scan(c);
        if (lane == 63) s_w[wave] = incl;
        const int grp = (int)blockIdx.x / SP_ROOT_GROUP;
        int part = 0;
        for (int b = threadIdx.x; b < grp; b += 256) part += grp_sum[b];
        if ((int)threadIdx.x < (int)blockIdx.x - grp * SP_ROOT_GROUP) part += blk_sum[grp * SP_ROOT_GROUP + threadIdx.x];
        part = wave_sum_i(part);
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        int run = s_red[0] + s_red[1] + s_red[2] + s_red[3] + incl - c;
        for (int k = 0; k < wave; ++k) run += s_w[k];
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1u;
            const uint32_t g = (uint32_t)w * 32u + (uint32_t)j;
            base[g] = run;
            run += sp_gran_count(bm, g);
        }
    }
}

// every site to its rank: sorted_idx[rank] = idx[i], perm[rank] = i, and (optionally) its feature row
__global__ __launch_bounds__(256) void k_sp_root_apply(const int4* __restrict__ idx, int cap, const int* __restrict__ n_dev, SpShape s,
                                                      SpRank r, int4* __restrict__ sorted_idx, int* __restrict__ perm,
                                                      const float* __restrict__ feat, int channels, float* __restrict__ feat_sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= live_rows(n_dev, cap)) return;
    const int4 c = idx[i];
    const int row = sp_rank_lookup(r, sp_key(s, c.x, c.y, c.z, c.w));
    if (row >= 0 && row < cap) {
        sorted_idx[row] = c;
        perm[row] = i;
        if (feat) {
            if (channels == 4) {
                reinterpret_cast<float4*>(feat_sorted)[row] = reinterpret_cast<const float4*>(feat)[i];
            } else {
                for (int k = 0; k < channels; ++k) feat_sorted[(size_t)row * channels + k] = feat[(size_t)i * channels + k];
            }
        }
    }
}

// ---- weight gradient of the sparse convolution (training) --------------------------------------------------------
//   dW[tap][ci][co] = sum over the rule pairs (i, o) of that tap of  x[i][ci] * g[o][co]
// One block per (chunk of SPW_CHUNK output rows, tap): the chunk's valid pairs of the tap are compacted into LDS (ballot + prefix
// count, in row order: the sum order is fixed), then taken 32 at a time: the paired rows of x and g are gathered into LDS and
// multiplied on v_mfma_f32_16x16x4_f32 with the PAIR index as the reduction dimension (A = x^T: lane (ci, pair), B = g: lane
// (pair, co); both read conflict-free along the channels).  A wave owns up to four of the ceil(Cin/16) x (Cout/16) output tiles.
// Every block writes its [Cin][Cout] partial sum; the caller adds the chunks (torch.sum: fixed order -> deterministic).
constexpr int SPW_CHUNK = 2048;     // output rows per block
constexpr int SPW_PAIRS = 32;       // pairs per MFMA stage

__global__ __launch_bounds__(256) void k_sp_wgrad(const float* __restrict__ x, const float* __restrict__ g,
                                                 const int* __restrict__ nbr, int n_out, const int* __restrict__ n_dev, int K,
                                                 int cin, int cout, float* __restrict__ partials /*[chunks][K][cin][cout]*/) {
    __shared__ int s_in[SPW_CHUNK], s_out[SPW_CHUNK];
    __shared__ int s_wcnt[4], s_total;
    __shared__ __attribute__((aligned(16))) float sX[SPW_PAIRS][64 + 16], sG[SPW_PAIRS][64 + 16];   // row stride 80: the two k-rows of a 32-lane read phase fall in disjoint banks
    const int chunk = blockIdx.x, tap = blockIdx.y;
    const int live = live_rows(n_dev, n_out);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lk = lane >> 4, ln = lane & 15;
    // 1. compaction of the chunk's pairs of this tap, in row order
    int base = 0;
    for (int r0 = 0; r0 < SPW_CHUNK; r0 += 256) {
        const int o = chunk * SPW_CHUNK + r0 + (int)threadIdx.x;
        const int i = o < live ? nbr[(size_t)o * K + tap] : -1;
        const unsigned long long bal = __ballot(i >= 0);
        if (lane == 0) s_wcnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base + __popcll(bal & lanemask_lt());
        for (int w = 0; w < wave; ++w) off += s_wcnt[w];
        if (i >= 0) { s_in[off] = i; s_out[off] = o; }
        base += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        __syncthreads();
    }
    const int n_pairs = base;
    // 2. tiles of this wave: tile id = mt * nct + nt over ceil(cin/16) x (cout/16); ids wave, wave + 4, ...
    const int mct = (cin + 15) / 16, nct = cout / 16, n_tiles = mct * nct;
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xq = cin / 4 > 0 ? (cin + 3) / 4 : 1, gq = cout / 4;     // 16-B pieces per row
    for (int p0 = 0; p0 < n_pairs; p0 += SPW_PAIRS) {
        // gather 32 rows of x and of g (zero rows past the end: they add nothing)
        for (int e = threadIdx.x; e < SPW_PAIRS * 16; e += 256) {
            const int p = e >> 4, q = e & 15;
            const bool ok = p0 + p < n_pairs;
            if (q < xq) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const float* src = x + (size_t)s_in[p0 + p] * cin + q * 4;
                    if (cin % 4 == 0) v = *reinterpret_cast<const float4*>(src);
                    else { v.x = src[0]; if (q * 4 + 1 < cin) v.y = src[1]; if (q * 4 + 2 < cin) v.z = src[2]; if (q * 4 + 3 < cin) v.w = src[3]; }
                }
                *reinterpret_cast<float4*>(&sX[p][q * 4]) = v;
            }
            if (q < gq) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float4*>(g + (size_t)s_out[p0 + p] * cout + q * 4);
                *reinterpret_cast<float4*>(&sG[p][q * 4]) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = wave + 4 * q;
            if (tile >= n_tiles) continue;
            const int mt = tile / nct, nt = tile - mt * nct;
            const bool a_ok = mt * 16 + ln < cin;
#pragma unroll
            for (int ks = 0; ks < SPW_PAIRS / 4; ++ks) {
                const float a = a_ok ? sX[ks * 4 + lk][mt * 16 + ln] : 0.f;
                const float b = sG[ks * 4 + lk][nt * 16 + ln];
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // 3. partial sums: D[row = lk*4 + r (ci)][col = ln (co)]
    float* dst = partials + ((size_t)chunk * K + tap) * cin * cout;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tile = wave + 4 * q;
        if (tile >= n_tiles) continue;
        const int mt = tile / nct, nt = tile - mt * nct;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = mt * 16 + lk * 4 + r;
            if (ci < cin) dst[(size_t)ci * cout + nt * 16 + ln] = acc[q][r];
        }
    }
}

// transposed rulebook for the backward pass: nbr_t[i][tap] = the output site that input site i feeds through `tap` (a given
// (input, tap) pair feeds exactly one output), -1 where there is none.  nbr_t must be pre-filled with -1.
__global__ __launch_bounds__(256) void k_sp_nbr_transpose(const int* __restrict__ nbr, int out_cap, const int* __restrict__ n_dev,
                                                         int K, int n_in, int* __restrict__ nbr_t) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)live_rows(n_dev, out_cap) * K) return;
    const int i = nbr[t];
    if (i >= 0 && i < n_in) nbr_t[(size_t)i * K + (int)(t % K)] = (int)(t / K);
}

// ---- the rulebook as pair tiles (round 6): what k_sp_tiles consumes, written directly --------------------------------------------------
// One wave per S output sites (lane = site, S / 64 sites per lane): the 27 neighbour rows come from the rank structure (the same
// lookups as k_sp_nbr_rank), every tap's live (input row, site) pairs are compacted by ballot and stored as whole 16-pair tiles into
// the wave's fixed-stride slot, the tile count and the tap of every tile into its header.  Against the [n_out][27] table: 6.2 of
// 27 taps are live on the full-resolution submanifold layers (2.5 of 27 on the first strided layer), so the rulebook WRITES
// ~2.6 KB instead of 6.9 KB per 64 sites, each convolution that uses it READS as much, and none of them compacts anything.
template <int S>
__global__ __launch_bounds__(256) void k_sp_nbr_tiles(const int4* __restrict__ out_idx, int out_cap,
                                                     const int* __restrict__ n_dev, SpConvGeom g, SpRank r, int in_cap,
                                                     const int* __restrict__ n_in_dev, uint32_t* __restrict__ tiles) {
    constexpr int CH = S / 64, LB = spt_lb(S);
    constexpr uint32_t PADW = (uint32_t)S;       // input row 0 -> the accumulator's trash row
    const int n_out = live_rows(n_dev, out_cap), n_in = live_rows(n_in_dev, in_cap);
    const int wave = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), l = threadIdx.x & 63;
    const int site0 = wave * S;
    if (site0 >= n_out) return;
    int v[CH][27];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int o = site0 + c * 64 + l;
        const bool live = o < n_out;
        const int4 cd = out_idx[live ? o : site0];
        const int z0 = cd.y * g.s[0] - g.p[0], y0 = cd.z * g.s[1] - g.p[1], x0 = cd.w * g.s[2] - g.p[2];
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int z = z0 + kz, y = y0 + ky;
                const bool line_ok = live && z >= 0 && z < g.in.D && y >= 0 && y < g.in.H;
                const uint32_t line = line_ok ? sp_key(g.in, cd.x, z, y, 0) : 0u;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int x = x0 + kx;
                    const int row = (line_ok && x >= 0 && x < g.in.W) ? sp_rank_lookup(r, line + (uint32_t)x) : -1;
                    v[c][(kz * 3 + ky) * 3 + kx] = row < n_in ? row : -1;
                }
            }
    }
    uint32_t* slot = tiles + (size_t)wave * spt_slot_words(S);
    uint32_t* list = slot + SPT_HDR_WORDS;
    uint8_t* taps = reinterpret_cast<uint8_t*>(slot + 1);
    int T = 0;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
        int n = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool h = v[c][t] >= 0;
            const unsigned long long have = __ballot(h);
            if (h) list[16 * T + n + __popcll(have & lanemask_lt())] = ((uint32_t)v[c][t] << LB) | (uint32_t)(c * 64 + l);
            n += __popcll(have);
        }
        const int nt = (n + 15) >> 4;
        if (l < 16 * nt - n) list[16 * T + n + l] = PADW;
        if (l < nt) taps[T + l] = (uint8_t)t;
        T += nt;
    }
    // whole groups of 4 tiles for the consumer's pipeline (all-padding tiles of tap 0), and one all-padding group behind them
    const int Tp = (T + 3) & ~3;
    for (int i = l; i < 16 * (Tp + 4 - T); i += 64) list[16 * T + i] = PADW;
    if (l < Tp + 4 - T) taps[T + l] = 0;
    if (l == 0) slot[0] = (uint32_t)Tp;
}

// pair tiles -> the [n_out][27] table (tests, bench's pair counts)
template <int S>
__global__ __launch_bounds__(256) void k_sp_tiles_to_nbr(const uint32_t* __restrict__ tiles, int out_cap,
                                                        const int* __restrict__ n_dev, int* __restrict__ nbr) {
    constexpr int LB = spt_lb(S);
    const int n_out = live_rows(n_dev, out_cap);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wave = (int)(blockIdx.x * (256 / 64) + w);
    const int site0 = wave * S;
    if (site0 >= n_out) return;
    __shared__ int s_tab[4][(S + 1) * 27];
    int* tab = s_tab[w];                          // wave-private: LDS operations of one wave execute in order
    const int rows = min(S, n_out - site0);
    for (int i = l; i < (S + 1) * 27; i += 64) tab[i] = -1;
    const uint32_t* slot = tiles + (size_t)wave * spt_slot_words(S);
    const int T = (int)slot[0];
    const uint8_t* taps = reinterpret_cast<const uint8_t*>(slot + 1);
    for (int i = l; i < 16 * T; i += 64) {
        const uint32_t e = slot[SPT_HDR_WORDS + i];
        const int m = (int)(e & (uint32_t)((1 << LB) - 1));
        if (m < S) tab[m * 27 + taps[i >> 4]] = (int)(e >> LB);
    }
    for (int i = l; i < rows * 27; i += 64) nbr[(size_t)site0 * 27 + i] = tab[i];
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

// Features of the sorted site set: out[i] = features[perm[i]] (perm from heal_sp_sort_sites; rows beyond the live count untouched).
extern "C" int heal_sp_gather_rows(const float* features, const int32_t* perm, int n, int channels, const int32_t* n_dev,
                                   float* out, void* stream) {
    HEAL_REQUIRE(n >= 0 && channels >= 1, "sp_gather_rows: bad shape");
    if (n == 0) return 0;
    HEAL_REQUIRE(features && perm && out, "sp_gather_rows: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (channels % 4 == 0 && (((uintptr_t)features | (uintptr_t)out) & 15) == 0) {
        const int pieces = channels / 4;
        k_sp_gather_rows<float4><<<(unsigned)(((long long)n * pieces + 255) / 256), 256, 0, s>>>(
            reinterpret_cast<const float4*>(features), perm, n, n_dev, pieces, reinterpret_cast<float4*>(out));
    } else {
        k_sp_gather_rows<float><<<(unsigned)(((long long)n * channels + 255) / 256), 256, 0, s>>>(features, perm, n, n_dev, channels, out);
    }
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
    HEAL_FILL(table_keys, 0xFF, table_cap * sizeof(uint32_t), s);
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
    if (n_in <= 0) { HEAL_FILL(n_out, 0, sizeof(int), s); return 0; }
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
    HEAL_FILL(okey, 0xFF, (size_t)cap * 4, s);
    const long long total = (long long)n_in * K;
    k_sp_candidates<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(reinterpret_cast<const int4*>(in_indices), n_in,
                                                                    n_in_dev, g, okey, cap - 1);
    k_sp_slot_flags<<<ceil_div((int)cap, 256), 256, 0, s>>>(okey, (int)cap, flag);
    if (scan_exclusive(flag, flag, (int)cap, n_out, cscratch, s)) return 1;
    // unique keys -> compact; pad the tail with 0xFFFFFFFF so that a fixed-size sort puts them last
    HEAL_FILL(keys[0], 0xFF, (size_t)out_cap * 4, s);
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


// ---- rank structure API --------------------------------------------------------------------------------------------------
static size_t rank_words(const SpShape& s) {
    const uint64_t cells = (uint64_t)s.B * s.D * s.H * s.W;
    return (size_t)((cells + SP_GRAN - 1) / SP_GRAN) * (SP_GRAN / 32);
}

extern "C" size_t heal_sp_rank_bytes(const int32_t* shape_host, int batch) {
    SpShape s;
    if (!shape_ok(shape_host, batch, s)) return 0;
    const size_t words = rank_words(s), gran = words / 8;
    return align_up(words * 4) + align_up(gran * 4) + align_up(scan_scratch_words((int64_t)gran) * 4) + 256;
}

// Active output sites of a strided sparse convolution through an occupancy bitmap of the OUTPUT grid (no hash, no sort):
// out_indices [out_cap,4] sorted by linear coordinate, n_out [1] device (not clamped; `overflow`, optional: a device word
// that receives max(overflow, n_out) whenever n_out exceeds out_cap -- sticky across graph replays).  `rank` (heal_sp_rank_bytes(out_shape,
// batch) bytes, 256-B aligned) is left holding the rank structure of the output site set: pass it to heal_sp_neighbors_rank
// when that set is the INPUT of a later layer.
extern "C" int heal_sp_out_sites_rank(const int32_t* in_indices, int n_in, const int32_t* ksize_host,
                                      const int32_t* stride_host, const int32_t* padding_host,
                                      const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                      int32_t* out_indices, int out_cap, int32_t* n_out, void* rank, size_t rank_bytes,
                                      const int32_t* n_in_dev, int32_t* overflow, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpConvGeom g;
    if (fill_geom(g, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch)) return 1;
    HEAL_REQUIRE(((uintptr_t)rank & 255) == 0, "sp_out_sites_rank: rank buffer must be 256-B aligned");
    const size_t words = rank_words(g.out), gran = words / 8;
    HEAL_REQUIRE(gran < (1ull << 31), "sp_out_sites_rank: grid too large");
    Arena a(rank, rank_bytes);
    uint32_t* bm = a.take<uint32_t>(words);
    int* base = a.take<int>(gran);
    int* scratch = a.take<int>(scan_scratch_words((int64_t)gran));
    HEAL_REQUIRE(a.ok(), "sp_out_sites_rank: rank buffer too small (%zu < %zu)", rank_bytes, a.off);
    HEAL_FILL(bm, 0, words * 4, s);
    if (n_in > 0)
        k_sp_bm_mark<<<ceil_div(n_in, 256), 256, 0, s>>>(reinterpret_cast<const int4*>(in_indices), n_in, n_in_dev, g, bm);
    k_sp_bm_count<<<(unsigned)((gran + 255) / 256), 256, 0, s>>>(reinterpret_cast<const uint4*>(bm), (int)gran, base);
    if (scan_exclusive(base, base, (int)gran, n_out, scratch, s)) return 1;
    k_sp_bm_emit<<<(unsigned)((words + 255) / 256), 256, 0, s>>>(bm, base, words, out_cap, g.out,
                                                                  reinterpret_cast<int4*>(out_indices), n_out, overflow);
    HEAL_LAUNCH_CHECK();
    return 0;
}

// nbr [n_out, K] through the rank structure of the input site set (built by heal_sp_out_sites_rank for `in_shape`); n_in /
// n_in_dev: rows of the input set (sites beyond its capacity were dropped when it was emitted: they are no neighbours).
// the root structure's buffer: bitmap | bases | level-1 bitmap (directory) + per-group site counts (one memset) | per-block and
// per-directory-word counts
struct SpRootWs {
    uint32_t* bm; int* base; uint32_t* l1; int* grp_sum; int* blk_sum; int* cnt;
    size_t words, gran, l1_words, clear_bytes;
    int blocks, groups;
};
static bool carve_root(const SpShape& s, void* rank, size_t rank_bytes, SpRootWs& w) {
    w.words = rank_words(s); w.gran = w.words / 8; w.l1_words = (w.gran + 31) / 32;
    w.blocks = (int)((w.l1_words + 255) / 256);
    Arena a(rank, rank_bytes);
    w.bm = a.take<uint32_t>(w.words);
    w.base = a.take<int>(w.gran);
    w.groups = (w.blocks + SP_ROOT_GROUP - 1) / SP_ROOT_GROUP;
    w.l1 = a.take<uint32_t>(w.l1_words);
    w.grp_sum = a.take<int>((size_t)w.groups);
    w.blk_sum = a.take<int>((size_t)w.blocks);
    w.cnt = a.take<int>(w.l1_words);
    w.clear_bytes = (size_t)((char*)w.blk_sum - (char*)w.l1);
    return a.ok();
}

static int neighbors_rank(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                          const int32_t* stride_host, const int32_t* padding_host,
                          const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                          const void* rank, size_t rank_bytes, int n_in, const int32_t* n_in_dev,
                          int32_t* nbr, const int32_t* n_out_dev, void* stream, bool root) {
    SpConvGeom g;
    if (fill_geom(g, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch)) return 1;
    if (n_out <= 0) return 0;
    const size_t words = rank_words(g.in), gran = words / 8;
    Arena a(const_cast<void*>(rank), rank_bytes);
    SpRank r;
    r.bm = a.take<uint32_t>(words);
    r.base = a.take<int>(gran);
    r.l1 = root ? a.take<uint32_t>((gran + 31) / 32) : nullptr;
    HEAL_REQUIRE(a.ok() && ((uintptr_t)rank & 255) == 0, "sp_neighbors_rank: bad rank buffer");
    const long long total = (long long)n_out * g.k[0] * g.k[1];
    k_sp_nbr_rank<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const int4*>(out_indices), n_out, n_out_dev, g, r, n_in, n_in_dev, nbr);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_sp_neighbors_rank(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                                      const int32_t* stride_host, const int32_t* padding_host,
                                      const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                      const void* rank, size_t rank_bytes, int n_in, const int32_t* n_in_dev,
                                      int32_t* nbr, const int32_t* n_out_dev, void* stream) {
    return neighbors_rank(out_indices, n_out, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch, rank,
                          rank_bytes, n_in, n_in_dev, nbr, n_out_dev, stream, false);
}

// the same query against the two-level structure heal_sp_root_rank leaves behind
extern "C" int heal_sp_neighbors_root(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                                      const int32_t* stride_host, const int32_t* padding_host,
                                      const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                      const void* rank, size_t rank_bytes, int n_in, const int32_t* n_in_dev,
                                      int32_t* nbr, const int32_t* n_out_dev, void* stream) {
    return neighbors_rank(out_indices, n_out, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch, rank,
                          rank_bytes, n_in, n_in_dev, nbr, n_out_dev, stream, true);
}

// 128-site slots, groups of 4 tiles and the timing-anatomy variants are measured alternatives (profiles/r06_k3_thin.json): they are built
// only into a HEAL_BUILD_EXPERIMENTAL=1 library, the shipped one carries what runs
#ifdef HEAL_BUILD_EXPERIMENTAL
static bool slot_sites_ok(int S) { return S == 64 || S == 128; }
#else
static bool slot_sites_ok(int S) { return S == 64; }
#endif

extern "C" size_t heal_sp_pair_tiles_words(int n_out, int slot_sites) {
    if (!slot_sites_ok(slot_sites)) return 0;
    return (size_t)ceil_div(n_out < 1 ? 1 : n_out, slot_sites) * spt_slot_words(slot_sites);
}

extern "C" int heal_sp_neighbor_tiles(const int32_t* out_indices, int n_out, const int32_t* ksize_host,
                                      const int32_t* stride_host, const int32_t* padding_host,
                                      const int32_t* in_shape_host, const int32_t* out_shape_host, int batch,
                                      const void* rank, size_t rank_bytes, int root, int n_in, const int32_t* n_in_dev,
                                      int slot_sites, uint32_t* tiles, const int32_t* n_out_dev, void* stream) {
    SpConvGeom g;
    if (fill_geom(g, ksize_host, stride_host, padding_host, in_shape_host, out_shape_host, batch)) return 1;
    HEAL_REQUIRE(g.k[0] == 3 && g.k[1] == 3 && g.k[2] == 3, "sp_neighbor_tiles: 3 x 3 x 3 kernels only");
    HEAL_REQUIRE(slot_sites_ok(slot_sites), "sp_neighbor_tiles: %d sites per slot (64; 128 in experimental builds)", slot_sites);
    HEAL_REQUIRE(n_in < (1 << 24), "sp_neighbor_tiles: %d input rows do not fit the pair word", n_in);
    if (n_out <= 0) return 0;
    const size_t words = rank_words(g.in), gran = words / 8;
    Arena a(const_cast<void*>(rank), rank_bytes);
    SpRank r;
    r.bm = a.take<uint32_t>(words);
    r.base = a.take<int>(gran);
    r.l1 = root ? a.take<uint32_t>((gran + 31) / 32) : nullptr;
    HEAL_REQUIRE(a.ok() && ((uintptr_t)rank & 255) == 0 && ((uintptr_t)tiles & 15) == 0, "sp_neighbor_tiles: bad rank / tile buffer");
    const int4* oi = reinterpret_cast<const int4*>(out_indices);
    if (slot_sites == 64)
        k_sp_nbr_tiles<64><<<ceil_div(n_out, 256), 256, 0, (hipStream_t)stream>>>(oi, n_out, n_out_dev, g, r, n_in, n_in_dev, tiles);
#ifdef HEAL_BUILD_EXPERIMENTAL
    else
        k_sp_nbr_tiles<128><<<ceil_div(n_out, 512), 256, 0, (hipStream_t)stream>>>(oi, n_out, n_out_dev, g, r, n_in, n_in_dev, tiles);
#endif
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_sp_tiles_to_neighbors(const uint32_t* tiles, int n_out, int slot_sites, const int32_t* n_out_dev, int32_t* nbr,
                                          void* stream) {
    HEAL_REQUIRE(slot_sites_ok(slot_sites), "sp_tiles_to_neighbors: %d sites per slot (64; 128 in experimental builds)", slot_sites);
    if (n_out <= 0) return 0;
    if (slot_sites == 64) k_sp_tiles_to_nbr<64><<<ceil_div(n_out, 256), 256, 0, (hipStream_t)stream>>>(tiles, n_out, n_out_dev, nbr);
#ifdef HEAL_BUILD_EXPERIMENTAL
    else k_sp_tiles_to_nbr<128><<<ceil_div(n_out, 512), 256, 0, (hipStream_t)stream>>>(tiles, n_out, n_out_dev, nbr);
#endif
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_sp_root_rank_bytes(const int32_t* shape_host, int batch) {
    SpShape s;
    if (!shape_ok(shape_host, batch, s)) return 0;
    const size_t words = rank_words(s), gran = words / 8, l1w = (gran + 31) / 32;
    const size_t blocks = (l1w + 255) / 256, groups = (blocks + SP_ROOT_GROUP - 1) / SP_ROOT_GROUP;
    return align_up(words * 4) + align_up(gran * 4) + 2 * align_up(l1w * 4) + align_up(groups * 4) + align_up(blocks * 4) + 256;
}

// Sort the voxel set by linear coordinate WITHOUT a sort and leave its rank structure in `rank` (heal_sp_root_rank_bytes(shape,
// batch) bytes, 256-B aligned, caller-owned, contents on entry irrelevant): sorted_indices[r] = indices[perm[r]], r = the number
// of sites with a smaller linear coordinate.  Same outputs as heal_sp_sort_sites (sites are unique); `features` [n, channels]
// (optional) are re-ordered on the way: sorted_features[r] = features[perm[r]] (= heal_sp_gather_rows).  heal_sp_neighbors_root
// then answers the neighbour queries of the layers that read this set.
extern "C" int heal_sp_root_rank(const int32_t* indices, int n, const int32_t* shape_host, int batch, int32_t* sorted_indices,
                                 int32_t* perm, const float* features, int channels, float* sorted_features, void* rank,
                                 size_t rank_bytes, const int32_t* n_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    SpShape sh;
    HEAL_REQUIRE(shape_ok(shape_host, batch, sh), "sp_root_rank: bad shape");
    HEAL_REQUIRE(((uintptr_t)rank & 255) == 0, "sp_root_rank: rank buffer must be 256-B aligned");
    HEAL_REQUIRE(features == nullptr || (channels >= 1 && sorted_features != nullptr), "sp_root_rank: bad feature arguments");
    HEAL_REQUIRE(features == nullptr || channels != 4 || (((uintptr_t)features | (uintptr_t)sorted_features) & 15) == 0,
                 "sp_root_rank: 4-channel feature rows must be 16-B aligned");
    if (n <= 0) return 0;
    SpRootWs w;
    HEAL_REQUIRE(carve_root(sh, rank, rank_bytes, w), "sp_root_rank: rank buffer too small (%zu bytes)", rank_bytes);
    HEAL_REQUIRE(w.gran < (1ull << 31), "sp_root_rank: grid too large");
    const int4* idx = reinterpret_cast<const int4*>(indices);
    const int nb = ceil_div(n, 256);
    HEAL_FILL(w.l1, 0, w.clear_bytes, s);
    k_sp_root_mark1<<<nb, 256, 0, s>>>(idx, n, n_dev, sh, w.l1, w.bm);
    k_sp_root_mark2<<<nb, 256, 0, s>>>(idx, n, n_dev, sh, w.bm);
    k_sp_root_count<0><<<w.blocks, 256, 0, s>>>(w.l1, w.bm, (int)w.l1_words, w.cnt, w.blk_sum, w.grp_sum, w.base);
    k_sp_root_count<1><<<w.blocks, 256, 0, s>>>(w.l1, w.bm, (int)w.l1_words, w.cnt, w.blk_sum, w.grp_sum, w.base);
    SpRank r;
    r.bm = w.bm; r.base = w.base; r.l1 = w.l1;
    k_sp_root_apply<<<nb, 256, 0, s>>>(idx, n, n_dev, sh, r, reinterpret_cast<int4*>(sorted_indices), perm, features, channels,
                                        sorted_features);
    HEAL_LAUNCH_CHECK();
    return 0;
}

// Backward rulebook (training): nbr_t [n_in, K] <- for every (o, tap) with i = nbr[o][tap] >= 0: nbr_t[i][tap] = o.  The gradient
// of a sparse convolution with respect to its input features is then the SAME gather-GEMM with the roles swapped:
//   d_in[i] = sum_tap W[tap] d_out[nbr_t[i][tap]]  =  heal_sp_conv(d_out, nbr_t, weight' = W[tap]^T ([Cout, Cin] per tap)).
extern "C" int heal_sp_transpose_neighbors(const int32_t* nbr, int n_out, int kernel_volume, int n_in, int32_t* nbr_t,
                                           const int32_t* n_out_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(kernel_volume >= 1 && kernel_volume <= 27 && n_in >= 0, "sp_transpose_neighbors: bad arguments");
    if (n_in > 0) HEAL_FILL(nbr_t, 0xFF, (size_t)n_in * kernel_volume * sizeof(int32_t), s);
    if (n_out <= 0 || n_in <= 0) return 0;
    const long long total = (long long)n_out * kernel_volume;
    k_sp_nbr_transpose<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(nbr, n_out, n_out_dev, kernel_volume, n_in, nbr_t);
    HEAL_LAUNCH_CHECK();
    return 0;
}

// out[o] = act( BN( sum_tap W[tap]^T in[nbr[o][tap]] ) ); weight [K][Cin][Cout].
extern "C" int heal_sp_conv(const float* feat_in, const int32_t* nbr, int n_out, int kernel_volume, int c_in,
                            int c_out, const float* weight, const float* weight_frag, const float* bn_scale,
                            const float* bn_shift, int relu, float* feat_out, const int32_t* n_out_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n_out <= 0) return 0;
    // Round 3: pair-compacted tiles (k_sp_conv2).  HEAL_SP_CONV=v1 keeps the round-2 kernel for A/B; HEAL_SP_M / HEAL_SP_TPS
    // select among the instantiated block shapes (tuning only).
    const char* mode = getenv("HEAL_SP_CONV");
    const int env_m = getenv("HEAL_SP_M") ? atoi(getenv("HEAL_SP_M")) : 0;
    if (weight_frag && !(mode && mode[0] == 'v' && mode[1] == '1') && kernel_volume <= 27) {
        const int env_tx = getenv("HEAL_SP_TPSX") ? atoi(getenv("HEAL_SP_TPSX")) : 0;
        const int env_db = getenv("HEAL_SP_DB") ? atoi(getenv("HEAL_SP_DB")) : -1;
        const int dbg = HEAL_DEBUG_ENV("HEAL_SP_DBG");
#define HEAL_SP2(CI, CO, MM, TT, DD, GG)                                                                                 \
    {                                                                                                                    \
        k_sp_conv2<CI, CO, MM, TT, DD, GG><<<ceil_div(n_out, MM), 256, 0, s>>>(feat_in, nbr, n_out, n_out_dev,           \
                                                                          kernel_volume, weight_frag, bn_scale, bn_shift, \
                                                                          relu, feat_out);                               \
        HEAL_LAUNCH_CHECK();                                                                                             \
        return 0;                                                                                                        \
    }
#define HEAL_SP2_CASE(CI, CO, TT, DEF_M, DEF_TX, DEF_DB)                                          \
    if (c_in == CI && c_out == CO) {                                                              \
        const int m_ = env_m ? env_m : DEF_M, tx_ = env_tx ? env_tx : DEF_TX;                     \
        const int db_ = env_db >= 0 ? env_db : DEF_DB;                                            \
        if (m_ == 64 && tx_ == 1 && db_ == 0) HEAL_SP2(CI, CO, 64, TT, 0, 0)                      \
        if (m_ == 64 && tx_ == 1 && db_ == 1) HEAL_SP2(CI, CO, 64, TT, 1, 0)                      \
        if (m_ == 64 && tx_ == 2 && db_ == 0) HEAL_SP2(CI, CO, 64, 2 * TT, 0, 0)                  \
        if (m_ == 64 && tx_ == 2 && db_ == 1) HEAL_SP2(CI, CO, 64, 2 * TT, 1, 0)                  \
        if (m_ == 128 && tx_ == 1 && db_ == 0) HEAL_SP2(CI, CO, 128, TT, 0, 0)                    \
        if (m_ == 128 && tx_ == 1 && db_ == 1) HEAL_SP2(CI, CO, 128, TT, 1, 0)                    \
        if (m_ == 128 && tx_ == 2 && db_ == 0) HEAL_SP2(CI, CO, 128, 2 * TT, 0, 0)                \
        if (m_ == 128 && tx_ == 2 && db_ == 1) HEAL_SP2(CI, CO, 128, 2 * TT, 1, 0)                \
    }
#define HEAL_SP2D(D) if (dbg == D && c_in == 64 && c_out == 64) HEAL_SP2(64, 64, 64, 2, 0, D)
        HEAL_SP2D(1) HEAL_SP2D(2) HEAL_SP2D(4) HEAL_SP2D(8) HEAL_SP2D(10) HEAL_SP2D(15) HEAL_SP2D(16)
#undef HEAL_SP2D
        // defaults from the sweep on the 8-agent SECOND encoder (profiles/r03_k3_sweep.json): 64 sites per block, 32 pairs x
        // 64 channels of gathered rows per stage, single-buffered gather tile (4 blocks per CU)
        HEAL_SP2_CASE(4, 16, 16, 64, 1, 0) HEAL_SP2_CASE(16, 16, 8, 64, 1, 0) HEAL_SP2_CASE(16, 32, 8, 64, 1, 0)
        HEAL_SP2_CASE(32, 32, 4, 64, 1, 0) HEAL_SP2_CASE(32, 64, 4, 64, 1, 0) HEAL_SP2_CASE(64, 64, 2, 64, 1, 0)
        HEAL_SP2_CASE(64, 128, 2, 64, 1, 0)
        // the transposed directions the gradient with respect to the input features runs through (heal_sp_transpose_neighbors)
        HEAL_SP2_CASE(32, 16, 4, 64, 1, 0) HEAL_SP2_CASE(64, 32, 2, 64, 1, 0) HEAL_SP2_CASE(128, 64, 1, 64, 1, 0)
#undef HEAL_SP2_CASE
#undef HEAL_SP2
    }
    HEAL_REQUIRE(weight != nullptr, "sp_conv: no reference-layout weight for the round-2 kernel (channels %d -> %d)", c_in, c_out);
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

extern "C" int heal_sp_conv_tiles_supported(int c_in, int c_out) {
    // (32 -> 32: 173 -> 166 us, and its rulebook launch is 18 us cheaper as tiles; 32 -> 64 measured no gain on 212 VGPRs: not built)
    return (c_in == 4 && c_out == 16) || (c_in == 16 && (c_out == 16 || c_out == 32)) || (c_in == 32 && c_out == 32);
}

extern "C" int heal_sp_conv_tiles(const float* feat_in, const uint32_t* tiles, int n_out, int slot_sites, int c_in, int c_out,
                                  const float* weight_frag, const float* bn_scale, const float* bn_shift, int relu,
                                  float* feat_out, const int32_t* n_out_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n_out <= 0) return 0;
    HEAL_REQUIRE(weight_frag != nullptr, "sp_conv_tiles: weight fragments (heal_sp_weight_fragments) are required");
    HEAL_REQUIRE(slot_sites_ok(slot_sites), "sp_conv_tiles: %d sites per slot (64; 128 in experimental builds)", slot_sites);
#ifdef HEAL_BUILD_EXPERIMENTAL
    const int dbg = HEAL_DEBUG_ENV("HEAL_SP_TILES_DBG");
    // tiles per pipeline group (HEAL_SP_TILES_D=4: A/B).  Measured on the three layers of config 5 (profiles/r06_k3_thin.json):
    // groups of 2 tiles 21.7 / 32.7 / 48.4 us on 36 / 64 / 88 VGPRs, groups of 4 22.2 / 34.9 / 52.5 us on 52 / 104 / 148
    const int four = getenv("HEAL_SP_TILES_D") && atoi(getenv("HEAL_SP_TILES_D")) == 4;
#endif
#define HEAL_SPTL2(CI, CO, SS, GG, DD)                                                                                   \
    {                                                                                                                    \
        k_sp_tiles<CI, CO, SS, GG, DD><<<ceil_div(n_out, SS), 64, 0, s>>>(feat_in, tiles, n_out, n_out_dev, weight_frag,  \
                                                                          bn_scale, bn_shift, relu, feat_out);           \
        HEAL_LAUNCH_CHECK();                                                                                             \
        return 0;                                                                                                        \
    }
#ifdef HEAL_BUILD_EXPERIMENTAL
#define HEAL_SPTL(CI, CO, SS, GG)                                                                                        \
    {                                                                                                                    \
        if (GG == 0 && four) HEAL_SPTL2(CI, CO, SS, 0, 4)                                                                \
        HEAL_SPTL2(CI, CO, SS, GG, 2)                                                                                    \
    }
#define HEAL_SPTT(CI, CO)                                                                                                \
    if (c_in == CI && c_out == CO) {                                                                                     \
        if (slot_sites == 128) HEAL_SPTL(CI, CO, 128, 0)                                                                 \
        if (dbg == 1) HEAL_SPTL(CI, CO, 64, 1)                                                                           \
        if (dbg == 2) HEAL_SPTL(CI, CO, 64, 2)                                                                           \
        if (dbg == 4) HEAL_SPTL(CI, CO, 64, 4)                                                                           \
        if (dbg == 7) HEAL_SPTL(CI, CO, 64, 7)                                                                           \
        if (dbg == 8) HEAL_SPTL(CI, CO, 64, 8)                                                                           \
        HEAL_SPTL(CI, CO, 64, 0)                                                                                         \
    }
#else
#define HEAL_SPTL(CI, CO, SS, GG) HEAL_SPTL2(CI, CO, SS, GG, 2)
#define HEAL_SPTT(CI, CO) if (c_in == CI && c_out == CO) HEAL_SPTL(CI, CO, 64, 0)
#endif
    HEAL_SPTT(4, 16) HEAL_SPTT(16, 16) HEAL_SPTT(16, 32) HEAL_SPTT(32, 32)
#undef HEAL_SPTT
#undef HEAL_SPTL
#undef HEAL_SPTL2
    return set_error("sp_conv_tiles: channel combination %d -> %d is not instantiated", c_in, c_out);
}

// Weight gradient of a sparse convolution: partials [heal_sp_wgrad_chunks(n_out)][K][Cin][Cout] (the caller sums over the chunks).
extern "C" int heal_sp_wgrad_chunks(int n_out) { return ceil_div(n_out < 1 ? 1 : n_out, SPW_CHUNK); }

extern "C" int heal_sp_wgrad(const float* feat_in, const float* grad_out, const int32_t* nbr, int n_out, int kernel_volume,
                             int c_in, int c_out, const int32_t* n_out_dev, float* partials, void* stream) {
    HEAL_REQUIRE(n_out >= 1 && kernel_volume >= 1 && kernel_volume <= 27, "sp_wgrad: bad sizes");
    HEAL_REQUIRE(c_in >= 1 && c_in <= 64 && c_out >= 16 && c_out <= 64 && c_out % 16 == 0,
                 "sp_wgrad: channels %d -> %d not supported (Cin <= 64, Cout in {16, 32, 48, 64})", c_in, c_out);
    HEAL_REQUIRE(feat_in && grad_out && nbr && partials, "sp_wgrad: null pointer");
    HEAL_REQUIRE((((uintptr_t)grad_out | (uintptr_t)feat_in) & 15) == 0, "sp_wgrad: 16-B alignment");
    k_sp_wgrad<<<dim3(heal_sp_wgrad_chunks(n_out), kernel_volume), 256, 0, (hipStream_t)stream>>>(
        feat_in, grad_out, nbr, n_out, n_out_dev, kernel_volume, c_in, c_out, partials);
    HEAL_LAUNCH_CHECK();
    return 0;
}

// weight [K][CIN][COUT] -> the B-fragment order of k_sp_conv2 (same size).  CIN >= 16: [tap][nb][j][lane = 16 g + ln][i] =
// W[tap][16 j + 4 g + i][16 nb + ln]; CIN < 16: [tap][nb][lane][kc] = W[tap][4 kc + g][16 nb + ln].
__global__ __launch_bounds__(256) void k_sp_weight_frag(const float* __restrict__ w, int K, int cin, int cout,
                                                       float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= K * cin * cout) return;
    const int per_tap = cin * cout, tap = t / per_tap;
    int r = t - tap * per_tap;
    const int nb = r / (cin * 16);
    r -= nb * cin * 16;
    int k, ln;
    if (cin >= 16) {
        const int j = r / 256, lane = (r >> 2) & 63, i = r & 3;
        k = 16 * j + 4 * (lane >> 4) + i;
        ln = lane & 15;
    } else {
        const int kc_n = cin / 4, lane = r / kc_n, kc = r - lane * kc_n;
        k = 4 * kc + (lane >> 4);
        ln = lane & 15;
    }
    out[t] = w[((size_t)tap * cin + k) * cout + nb * 16 + ln];
}

extern "C" int heal_sp_weight_fragments(const float* weight, int kernel_volume, int c_in, int c_out, float* out,
                                        void* stream) {
    HEAL_REQUIRE(kernel_volume >= 1 && c_in >= 4 && c_in % 4 == 0 && (c_in < 16 || c_in % 16 == 0) && c_out % 16 == 0,
                 "sp_weight_fragments: unsupported channels %d -> %d", c_in, c_out);
    const int n = kernel_volume * c_in * c_out;
    k_sp_weight_frag<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(weight, kernel_volume, c_in, c_out, out);
    HEAL_LAUNCH_CHECK();
    return 0;
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
    HEAL_FILL(cell_map, 0xFF, cells * 4, s);
    if (n > 0)
        k_sp_fill_map<<<ceil_div(n, 256), 256, 0, s>>>(reinterpret_cast<const int4*>(indices), n, n_dev, sh, cell_map);
    const int cells4 = sh.H * sh.W / 4;
    dim3 grid(ceil_div(cells4, 256), ceil_div(channels, 16), batch * sh.D);
    k_sp_dense<<<grid, 256, 0, s>>>(reinterpret_cast<const int4*>(cell_map), features, cells4, channels, sh.D,
                                    reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
}
