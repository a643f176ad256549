// Token-major fp32 GEMM with fused prologue / epilogue for the V2X-ViT fusion transformer, plus its two small companions
// (LayerNorm statistics, split-attention weights).
//
// Reference call sites: opencood/models/sub_modules/base_transformer.py:7-40 (PreNorm = LayerNorm + fn, FeedForward = Linear,
// GELU, Linear), hmsa.py:38-151 (per-type q/k/v/a Linear layers around the per-pixel agent attention), mswin.py:46-122 (to_qkv /
// to_out around the window attention), split_attn.py:6-62 (global average -> fc1 -> LayerNorm -> ReLU -> fc2 -> softmax over the
// three window branches -> weighted sum), v2xvit_basic.py:158-192 (the residual adds between them).  Round 2 ran these as
// hipBLASLt GEMMs (7.5 ms of the 33 ms step of BASELINE config 5) with ATen LayerNorm / GELU / add / mul / mean / cat kernels
// between them (another 5 ms, each streaming the 134 MB token tensor).
//
// heal_linear: Y = act(norm(X) W^T * colscale + bias) + residual for X [T, K] row-major (tokens x channels), W [N, K] (the
// nn.Linear layout: y = x W^T, every output column is one contiguous weight row).
//   * 128 x 128 output tile per block, 4 waves x (64 x 64) on v_mfma_f32_32x32x2_f32 (16 accumulator registers per 32 x 32
//     block: 64 per lane).  One MFMA is 64 cycles for ONE A and ONE B float per lane, so the matrix pipe is fed by two
//     ds_read_b128 per operand per 16 k-steps; K chunks of 32 through double-buffered LDS (73 KB: two blocks per CU).
//   * both operands are staged [row][k] (k contiguous) with a 36-float row stride: the b128 fragment reads of the 16-lane
//     groups fall on distinct bank slots.  Lane half h multiplies k = 16 h + s at step s -- a permutation of the reduction
//     order that both operands share.
//   * prologue, applied while a tile travels global -> registers -> LDS: (x - mean[t]) * rstd[t] (PreNorm's LayerNorm; its
//     gamma / beta are folded into W / bias by the caller), and for the split-attention merge W[n][k] * colscale[group(t)][k /
//     part][n]: the three to_out projections and the softmax-weighted sum over the window branches are ONE K = 3 x 256 GEMM
//     whose weight rows are scaled per agent.
//   * epilogue: bias (optionally per group of tokens), exact-erf GELU, residual add, and an output addressing that writes
//     column parts to separate buffers (q | k | v) and rows through a [outer, inner] -> [inner, outer] transposition (the
//     per-pixel agent attention wants [pixel, agent, C], the token tensor is [agent, pixel, C]).
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct LinearArgs {
    const float* X; int lda;          // [T, K] rows lda apart
    int a_part_cols; long long a_part_stride;  // > 0: channel block k / a_part_cols of a token lives in X + block * a_part_stride
    const float2* stats;              // [T] (mean, rstd) or NULL
    const float* W;                   // [N, K]
    const float* bias;                // [groups, N] or [N] or NULL
    const float* colscale;            // [groups, K / cs_part, N] or NULL: W[n][k] *= colscale[g][k / cs_part][n]
    const float* res; int ldr;        // [T_out rows, N] residual (indexed like the output) or NULL
    float* out; int ldo;              // output rows ldo apart
    int T, N, K;
    int group_rows;                   // tokens per group (bias / colscale groups), 0 = one group
    int bias_groups;                  // 1: bias is per group
    int cs_part;                      // k extent of one colscale part
    int map_inner, map_outer;         // out row of token t = (t % inner) * outer + t / inner  (inner = 0: identity)
    int part_cols; long long part_stride;  // column c goes to out + (c / part_cols) * part_stride, column c % part_cols
    int act;                          // 0 none, 1 gelu (erf), 2 relu
};

constexpr int LIN_BM = 128, LIN_BN = 128, LIN_BK = 32, LIN_RS = 36;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256, 2) void k_linear(const LinearArgs a) {
    __shared__ __attribute__((aligned(16))) float s_all[2 * LIN_BM * LIN_RS + 2 * LIN_BN * LIN_RS];
    float (*sA)[LIN_BM * LIN_RS] = reinterpret_cast<float (*)[LIN_BM * LIN_RS]>(s_all);
    float (*sB)[LIN_BN * LIN_RS] = reinterpret_cast<float (*)[LIN_BN * LIN_RS]>(s_all + 2 * LIN_BM * LIN_RS);
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = l & 31, h = l >> 5;
    // blocks that share an A tile (same token tile, different column tiles) are neighbours in dispatch order
    const int n_tiles = a.N / LIN_BN;
    const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
    const int t0 = tile_m * LIN_BM, n0 = tile_n * LIN_BN;

    // staging role: 4 x 16 B of the A tile and 4 x 16 B of the B tile per chunk; row r = (tid + 256 i) / 8, 16-B column c4
    int s_row[4], s_c4[4];
    float s_mean[4], s_rstd[4];
    const float* a_src[4];
    const float* b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        s_row[i] = c >> 3;
        s_c4[i] = c & 7;
        const int t = min(t0 + s_row[i], a.T - 1);
        a_src[i] = a.X + (size_t)t * a.lda + s_c4[i] * 4;
        b_src[i] = a.W + (size_t)(n0 + s_row[i]) * a.K + s_c4[i] * 4;
        s_mean[i] = 0.f; s_rstd[i] = 1.f;
        if (a.stats) { const float2 st = a.stats[t]; s_mean[i] = st.x; s_rstd[i] = st.y; }
    }
    const int group = a.group_rows > 0 ? t0 / a.group_rows : 0;
    const float* cs_base = a.colscale ? a.colscale + (size_t)group * (a.K / a.cs_part) * a.N : nullptr;

    // reduction index -> offset inside a token's row (channel blocks of several tensors: see a_part_cols)
    auto a_off = [&](int k0) -> long long {
        if (a.a_part_cols <= 0) return k0;
        const int blk = k0 / a.a_part_cols;
        return (long long)blk * a.a_part_stride + (k0 - blk * a.a_part_cols);
    };
    float4 ra[4], rb[4];
#define HEAL_LIN_LOAD(k0_)                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                     \
        ra[i] = *reinterpret_cast<const float4*>(a_src[i] + a_off(k0_));                \
        rb[i] = *reinterpret_cast<const float4*>(b_src[i] + (k0_));                     \
    }
#define HEAL_LIN_STORE(buf_, k0_)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        float4 x = ra[i];                                                                                              \
        x.x = (x.x - s_mean[i]) * s_rstd[i]; x.y = (x.y - s_mean[i]) * s_rstd[i];                                      \
        x.z = (x.z - s_mean[i]) * s_rstd[i]; x.w = (x.w - s_mean[i]) * s_rstd[i];                                      \
        *reinterpret_cast<float4*>(&sA[buf_][s_row[i] * LIN_RS + s_c4[i] * 4]) = x;                                    \
        float4 w = rb[i];                                                                                              \
        if (cs_base) {                                                                                                 \
            const float sc = cs_base[(size_t)((k0_) / a.cs_part) * a.N + n0 + s_row[i]];                               \
            w.x *= sc; w.y *= sc; w.z *= sc; w.w *= sc;                                                                \
        }                                                                                                              \
        *reinterpret_cast<float4*>(&sB[buf_][s_row[i] * LIN_RS + s_c4[i] * 4]) = w;                                    \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // Measured and dropped: a static s_setprio for every other block (to de-phase the two blocks that share a CU): no change;
    // one barrier per chunk placed in the middle of its MFMAs with half-chunk fragment double buffering (every LDS read then
    // has 2048 cycles of MFMAs to land): 104-108 vs 104-110 TFLOP/s at 218 instead of 184 registers -- the chunk loop is not
    // where the remaining 20-25 % to the fp32 matrix roof is lost at this clock.
    const int n_chunks = a.K / LIN_BK;
    HEAL_LIN_LOAD(0)
    HEAL_LIN_STORE(0, 0)
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) { HEAL_LIN_LOAD((c + 1) * LIN_BK) }   // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        float af[2][16], bf[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* p = &sA[buf][(wm * 64 + m * 32 + li) * LIN_RS + 16 * h];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                af[m][4 * q] = v.x; af[m][4 * q + 1] = v.y; af[m][4 * q + 2] = v.z; af[m][4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* p = &sB[buf][(wn * 64 + n * 32 + li) * LIN_RS + 16 * h];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                bf[n][4 * q] = v.x; bf[n][4 * q + 1] = v.y; bf[n][4 * q + 2] = v.z; bf[n][4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][s], bf[n][s], acc[m][n], 0, 0, 0);
        if (c + 1 < n_chunks) { HEAL_LIN_STORE(buf ^ 1, (c + 1) * LIN_BK) }
        __syncthreads();
    }

#undef HEAL_LIN_LOAD
#undef HEAL_LIN_STORE
    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    // The accumulators go through LDS once (C/D layout of 32x32: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)),
    // so that every thread then owns 16-B pieces of output rows: bias / residual / output move as 16-B accesses (16 per thread
    // instead of 64 scalar ones), and the row / part addressing is computed once per piece.
    constexpr int CS = LIN_BN + 4;
    float* sC = s_all;       // 128 x 132 floats = 67.6 KB of the 73.7 KB the two operand rings occupy (contiguous arrays)
    static_assert(sizeof(s_all) >= (size_t)LIN_BM * CS * 4, "epilogue tile must fit the operand rings");
    // this thread's 16-B column is fixed: 32 column pieces per row, 8 rows per pass
    const int c4 = tid & 31, col = n0 + c4 * 4;
    // row map (a tile never straddles an `inner` boundary: inner is a multiple of 128)
    const long long q_ = a.map_inner > 0 ? t0 / a.map_inner : 0, rem_ = a.map_inner > 0 ? t0 - q_ * a.map_inner : t0;
    // The residual rows are requested HERE, before the accumulators go through LDS: 16 loads in flight under the transposition
    // instead of four exposed round trips inside the store loop (the accumulator registers are free from the LDS writes on).
    float4 rres[16];
    if (a.res) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min((tid >> 5) + 8 * i, a.T - 1 - t0);     // clamped: rows past T are never stored
            const size_t orow = a.map_inner > 0 ? (size_t)(rem_ + row) * a.map_outer + (size_t)q_ : (size_t)(t0 + row);
            rres[i] = *reinterpret_cast<const float4*>(a.res + orow * a.ldr + col);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CS + wn * 64 + n * 32 + li] = acc[m][n][r];
    __syncthreads();
    const float* bias = a.bias ? a.bias + (a.bias_groups ? (size_t)group * a.N : 0) : nullptr;
    const int part = col / a.part_cols, pc = col - part * a.part_cols;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + col);
    float* obase = a.out + (size_t)part * a.part_stride + pc;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (tid >> 5) + 8 * i;
        if (t0 + row >= a.T) break;
        const size_t orow = a.map_inner > 0 ? (size_t)(rem_ + row) * a.map_outer + (size_t)q_ : (size_t)(t0 + row);
        float4 v = *reinterpret_cast<const float4*>(&sC[row * CS + c4 * 4]);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (a.act == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        else if (a.act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (a.res) { v.x += rres[i].x; v.y += rres[i].y; v.z += rres[i].z; v.w += rres[i].w; }
        *reinterpret_cast<float4*>(obase + orow * a.ldo) = v;
    }
}

// ---- LayerNorm statistics: one wave per token (C <= 512), four tokens per wave at C == 256 ------------------------------------
__global__ __launch_bounds__(256) void k_ln_stats(const float* __restrict__ x, int T, int C, float eps,
                                                 float2* __restrict__ stats) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (t >= T) return;
    const float* row = x + (size_t)t * C;
    float v[8];
    int nv = 0;
    float s = 0.f;
    for (int c = l * 4; c < C; c += 256, nv += 4) {   // C <= 512
        const float4 q = *reinterpret_cast<const float4*>(row + c);
        v[nv] = q.x; v[nv + 1] = q.y; v[nv + 2] = q.z; v[nv + 3] = q.w;
        s += (q.x + q.y) + (q.z + q.w);
    }
    const float mean = wave_sum(s) / (float)C;
    float d = 0.f;
    for (int i = 0; i < nv; ++i) d += (v[i] - mean) * (v[i] - mean);
    const float var = wave_sum(d) / (float)C;
    if (l == 0) stats[t] = make_float2(mean, 1.0f / sqrtf(var + eps));
}

// C == 256: a wave takes FOUR tokens (its four 1-KB rows requested together; the same per-token arithmetic order as above, so
// the statistics are bit-identical) -- one row per wave left the loads of a CU too shallow for HBM (2.9 TB/s).
__global__ __launch_bounds__(256) void k_ln_stats256(const float* __restrict__ x, int T, float eps, float2* __restrict__ stats) {
    const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4, l = threadIdx.x & 63;
    if (t0 >= T) return;
    float4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const float4*>(x + (size_t)min(t0 + i, T - 1) * 256 + l * 4);
    float mean[4], var[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mean[i] = wave_sum((q[i].x + q[i].y) + (q[i].z + q[i].w)) / 256.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float d = 0.f;
        d += (q[i].x - mean[i]) * (q[i].x - mean[i]);
        d += (q[i].y - mean[i]) * (q[i].y - mean[i]);
        d += (q[i].z - mean[i]) * (q[i].z - mean[i]);
        d += (q[i].w - mean[i]) * (q[i].w - mean[i]);
        var[i] = wave_sum(d) / 256.0f;
    }
    if (l < 4 && t0 + l < T) {
        const float m = l == 0 ? mean[0] : l == 1 ? mean[1] : l == 2 ? mean[2] : mean[3];
        const float vv = l == 0 ? var[0] : l == 1 ? var[1] : l == 2 ? var[2] : var[3];
        stats[t0 + l] = make_float2(m, 1.0f / sqrtf(vv + eps));
    }
}

// ---- split attention weights (split_attn.py:43-62) ------------------------------------------------------------------------
// colsum[g][part][c]: sums over the group's tokens of the three window-attention outputs (BEFORE their to_out projections;
// the projections are linear, so the average of the projected branches is the projection of the averages).
//   gap = sum_part (mean_part W_out[part]^T + b_out[part]);  a = fc2(relu(LN(fc1(gap))));  softmax over the 3 parts per channel
// -> scale [g][3][C] (the column scales of the merged to_out GEMM) and bias [g][C] = sum_part scale * b_out[part].
// A row is read as C / 4 lanes x 16 B; the block's 256 threads cover 256 / (C / 4) rows per step and keep four steps in flight
// (one float per thread and four rows in flight left this 400-MB read at 4.3 TB/s); the row groups are then summed through LDS in
// a fixed order, so the sums do not depend on the execution order.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, long long part_stride, int rows_per_group, int C,
                                               int chunk, float* __restrict__ out /*[g][parts][chunks][C]*/) {
    __shared__ float4 s_part[256];
    const int g = blockIdx.z, part = blockIdx.y;
    const int r0 = blockIdx.x * chunk, r1 = min(r0 + chunk, rows_per_group);
    const int lpr = C >> 2, rpg = 256 / lpr;           // lanes per row (16 / 32 / 48 / 64), rows per step (16 / 8 / 5 / 4; at C = 192 sixteen threads idle)
    const int c4 = threadIdx.x % lpr, rg = threadIdx.x / lpr;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (rg < rpg) {
        const float* base = x + (size_t)part * part_stride + (size_t)g * rows_per_group * C + c4 * 4;
        int r = r0 + rg;
        for (; r + 3 * rpg < r1; r += 4 * rpg) {
            const float4 a = *reinterpret_cast<const float4*>(base + (size_t)r * C);
            const float4 b = *reinterpret_cast<const float4*>(base + (size_t)(r + rpg) * C);
            const float4 c = *reinterpret_cast<const float4*>(base + (size_t)(r + 2 * rpg) * C);
            const float4 d = *reinterpret_cast<const float4*>(base + (size_t)(r + 3 * rpg) * C);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; r < r1; r += rpg) {
            const float4 a = *reinterpret_cast<const float4*>(base + (size_t)r * C);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    s_part[threadIdx.x] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                      (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (threadIdx.x < lpr) {
        float4 t = s_part[threadIdx.x];
        for (int k = 1; k < rpg; ++k) {
            const float4 u = s_part[k * lpr + threadIdx.x];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(out + (((size_t)g * gridDim.y + part) * gridDim.x + blockIdx.x) * C + threadIdx.x * 4) = t;
    }
}

// out[row] = (bias ? bias[row] : 0) + w[row][0..C) . vec[(row / C) * vstride ..], rows strided over the block's 16 waves: a wave
// reads a weight row as 64 x 16 B (coalesced; one thread per row would stride the lanes by a whole row) and reduces across
// lanes.  C <= 256, C % 64 == 0, rows % 4 == 0.  The kernel runs on ONE block per group and is pure latency: a wave requests 16
// rows (four steps of four) with unconditional, clamped loads BEFORE it reduces any of them -- the round-3 version (4 waves, one
// step in flight, predicated loads) spent 145 us on seven 256 x 256 matrix-vector products.
__device__ __forceinline__ void block_matvec(const float* __restrict__ w, const float* __restrict__ bias,
                                             const float* __restrict__ vec /*LDS*/, int vstride, float* __restrict__ out /*LDS*/,
                                             int rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool on = lane * 4 < C;
    const int lc = on ? lane * 4 : 0;
    for (int rb = 0; rb < rows; rb += nw * 16) {
        float4 q[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = min(rb + (it * nw + wave) * 4 + jj, rows - 1);
                q[it][jj] = *reinterpret_cast<const float4*>(w + (size_t)r * C + lc);
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r0 = rb + (it * nw + wave) * 4;
            if (r0 >= rows) break;                          // wave-uniform
            float4 v = *reinterpret_cast<const float4*>(vec + (r0 / C) * vstride + lc);
            if (!on) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float d[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 x = q[it][jj];
                d[jj] = wave_sum(fmaf(x.w, v.w, fmaf(x.z, v.z, fmaf(x.y, v.y, x.x * v.x))));
            }
            if (lane < 4) {
                const float t = lane == 0 ? d[0] : lane == 1 ? d[1] : lane == 2 ? d[2] : d[3];
                out[r0 + lane] = t + (bias ? bias[r0 + lane] : 0.f);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_split_weights(const float* __restrict__ colsum /*[parts][g][3][chunks][C]*/, int chunks,
                                                       int n_parts, float inv_rows,
                                                       const float* __restrict__ w_out /*[3][C][C] (nn.Linear [N][K])*/,
                                                       const float* __restrict__ b_out /*[3][C]*/,
                                                       const float* __restrict__ fc1 /*[C][C]*/, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float eps,
                                                       const float* __restrict__ fc2 /*[3C][C]*/, int C,
                                                       float* __restrict__ scale /*[g][3][C]*/, float* __restrict__ bias /*[g][C]*/) {
    __shared__ __attribute__((aligned(16))) float s_mean[3 * 256], s_gap[256], s_h[256], s_t[3 * 256];
    __shared__ float s_red[8];
    const int g = blockIdx.x, c = threadIdx.x;   // 1024 threads; channel-indexed steps use the first C <= 256 of them
    const bool ch = c < C;
    // the partial sums of a (group, branch, channel) are added in the order of the token chunks they cover: part-major (a part
    // = the stripe of one rank when the tokens of a group are spread over ranks, dist.py), then chunk -- with whole 512-token
    // chunks per stripe this is the order of the unsharded call, whatever the number of parts
    const size_t part_stride = (size_t)gridDim.x * 3 * chunks * C;
    const int n_sums = n_parts * chunks;
    for (int i = c; i < 3 * C; i += blockDim.x) {
        const float* q0 = colsum + ((size_t)g * 3 + i / C) * chunks * C + (i % C);
        auto q = [&](int k) { return q0[(size_t)(k / chunks) * part_stride + (size_t)(k % chunks) * C]; };
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;     // four interleaved partial sums: a fixed order, loads in flight together
        int k = 0;
        for (; k + 3 < n_sums; k += 4) { t0 += q(k); t1 += q(k + 1); t2 += q(k + 2); t3 += q(k + 3); }
        for (; k < n_sums; ++k) t0 += q(k);
        s_mean[i] = ((t0 + t1) + (t2 + t3)) * inv_rows;
    }
    __syncthreads();
    block_matvec(w_out, b_out, s_mean, C, s_t, 3 * C, C);       // the three to_out projections: row block p reads mean p
    __syncthreads();
    if (ch) s_gap[c] = (s_t[c] + s_t[C + c]) + s_t[2 * C + c];
    __syncthreads();
    block_matvec(fc1, nullptr, s_gap, 0, s_t, C, C);
    __syncthreads();
    const float hval = ch ? s_t[c] : 0.f;
    // LayerNorm over the C values of the block (the waves past C hold zeros and are not summed)
    float sm = wave_sum(hval);
    if (ch && (c & 63) == 0) s_red[c >> 6] = sm;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (C + 63) / 64; ++i) tot += s_red[i];
    const float mean = tot / (float)C;
    __syncthreads();
    const float dv = (hval - mean) * (hval - mean);
    sm = wave_sum(dv);
    if (ch && (c & 63) == 0) s_red[c >> 6] = sm;
    __syncthreads();
    tot = 0.f;
    for (int i = 0; i < (C + 63) / 64; ++i) tot += s_red[i];
    const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
    if (ch) s_h[c] = fmaxf((hval - mean) * rstd * ln_g[c] + ln_b[c], 0.f);
    __syncthreads();
    block_matvec(fc2, nullptr, s_h, 0, s_t, 3 * C, C);
    __syncthreads();
    if (!ch) return;
    const float lg[3] = {s_t[c], s_t[C + c], s_t[2 * C + c]};
    const float mx = fmaxf(lg[0], fmaxf(lg[1], lg[2]));
    const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), e2 = expf(lg[2] - mx);
    const float inv = 1.0f / (e0 + e1 + e2);
    const float a0 = e0 * inv, a1 = e1 * inv, a2 = e2 * inv;
    scale[((size_t)g * 3 + 0) * C + c] = a0;
    scale[((size_t)g * 3 + 1) * C + c] = a1;
    scale[((size_t)g * 3 + 2) * C + c] = a2;
    bias[(size_t)g * C + c] = a0 * b_out[c] + a1 * b_out[C + c] + a2 * b_out[2 * C + c];
}

}  // namespace heal

using namespace heal;

extern "C" int heal_ln_stats(const float* x, int n_tokens, int channels, float eps, float* stats, void* stream) {
    HEAL_REQUIRE(channels % 4 == 0 && channels <= 512, "ln_stats: channels must be a multiple of 4, <= 512");
    if (n_tokens <= 0) return 0;
    if (channels == 256)
        k_ln_stats256<<<ceil_div(n_tokens, 16), 256, 0, (hipStream_t)stream>>>(x, n_tokens, eps, reinterpret_cast<float2*>(stats));
    else
        k_ln_stats<<<ceil_div(n_tokens, 4), 256, 0, (hipStream_t)stream>>>(x, n_tokens, channels, eps,
                                                                          reinterpret_cast<float2*>(stats));
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_linear(const float* x, int lda, int x_part_cols, long long x_part_stride, const float* ln_stats,
                           const float* weight, const float* bias,
                           int bias_per_group, const float* colscale, int colscale_part, int group_rows,
                           const float* residual, int ldr, float* out, int ldo, int n_tokens, int n_out, int n_in,
                           int map_inner, int map_outer, int part_cols, long long part_stride, int act, void* stream) {
    HEAL_REQUIRE(n_out % LIN_BN == 0 && n_in % LIN_BK == 0 && n_in >= LIN_BK, "linear: N must be a multiple of 128, K of 32");
    HEAL_REQUIRE(lda % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)weight & 15) == 0, "linear: 16-B alignment");
    HEAL_REQUIRE(group_rows == 0 || group_rows % LIN_BM == 0, "linear: a group must be whole 128-token tiles");
    HEAL_REQUIRE(!colscale || (colscale_part > 0 && colscale_part % LIN_BK == 0 && n_in % colscale_part == 0),
                 "linear: bad colscale part");
    HEAL_REQUIRE(map_inner == 0 || (map_inner > 0 && map_outer > 0 && (long long)map_inner * map_outer == n_tokens &&
                                    map_inner % LIN_BM == 0),
                 "linear: row map must cover the tokens, inner a multiple of 128");
    HEAL_REQUIRE(n_out % 4 == 0 && (part_cols <= 0 || part_cols % 4 == 0) && ldo % 4 == 0 && (!residual || ldr % 4 == 0),
                 "linear: 16-B output pieces");
    HEAL_REQUIRE(act >= 0 && act <= 2, "linear: act in {0,1,2}");
    if (n_tokens <= 0) return 0;
    LinearArgs a;
    HEAL_REQUIRE(x_part_cols == 0 || (x_part_cols % LIN_BK == 0 && n_in % x_part_cols == 0 && !ln_stats),
                 "linear: x parts must be whole 32-channel chunks (and carry no LayerNorm)");
    a.a_part_cols = x_part_cols; a.a_part_stride = x_part_stride;
    a.X = x; a.lda = lda; a.stats = reinterpret_cast<const float2*>(ln_stats); a.W = weight; a.bias = bias;
    a.colscale = colscale; a.res = residual; a.ldr = ldr; a.out = out; a.ldo = ldo; a.T = n_tokens; a.N = n_out; a.K = n_in;
    a.group_rows = group_rows; a.bias_groups = bias_per_group; a.cs_part = colscale_part > 0 ? colscale_part : n_in;
    a.map_inner = map_inner; a.map_outer = map_outer;
    a.part_cols = part_cols > 0 ? part_cols : n_out; a.part_stride = part_stride; a.act = act;
    const int tiles = ceil_div(n_tokens, LIN_BM) * (n_out / LIN_BN);
    HEAL_LAUNCH_EV(k_linear, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_split_attn_workspace(int groups, int rows_per_group, int channels) {
    return (size_t)groups * 3 * ceil_div(rows_per_group, 512) * channels * sizeof(float) + 256;
}

static int split_attn_colsum(const float* branches, long long part_stride, int groups, int rows_per_group, int channels,
                            float* colsum, hipStream_t s) {
    HEAL_REQUIRE(channels <= 256 && channels % 64 == 0, "split_attn_weights: channels must be 64, 128, 192 or 256");
    HEAL_REQUIRE(part_stride % 4 == 0 && ((uintptr_t)branches & 15) == 0 && ((uintptr_t)colsum & 15) == 0,
                 "split_attn_weights: 16-B alignment");
    const int chunk = 512, chunks = ceil_div(rows_per_group, chunk);
    dim3 grid(chunks, 3, groups);
    k_colsum<<<grid, 256, 0, s>>>(branches, part_stride, rows_per_group, channels, chunk, colsum);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_split_attn_colsum(const float* branches, long long part_stride, int groups, int rows_per_group, int channels,
                                      float* colsum, void* stream) {
    return split_attn_colsum(branches, part_stride, groups, rows_per_group, channels, colsum, (hipStream_t)stream);
}

extern "C" int heal_split_attn_weights_from_colsum(const float* colsum, int n_parts, int groups, int rows_per_part, int channels,
                                                   const float* w_out, const float* b_out, const float* fc1,
                                                   const float* ln_gamma, const float* ln_beta, float eps, const float* fc2,
                                                   float* scale, float* bias, void* stream) {
    HEAL_REQUIRE(channels <= 256 && channels % 64 == 0, "split_attn_weights: channels must be 64, 128, 192 or 256");
    HEAL_REQUIRE(n_parts >= 1 && rows_per_part >= 1, "split_attn_weights: n_parts, rows_per_part >= 1");
    const int chunks = ceil_div(rows_per_part, 512);
    k_split_weights<<<groups, 1024, 0, (hipStream_t)stream>>>(colsum, chunks, n_parts, 1.0f / ((float)rows_per_part * (float)n_parts),
                                                            w_out, b_out, fc1, ln_gamma, ln_beta, eps, fc2, channels, scale, bias);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_split_attn_weights(const float* branches, long long part_stride, int groups, int rows_per_group,
                                       int channels, const float* w_out, const float* b_out, const float* fc1,
                                       const float* ln_gamma, const float* ln_beta, float eps, const float* fc2,
                                       float* colsum_ws, float* scale, float* bias, void* stream) {
    if (int rc = split_attn_colsum(branches, part_stride, groups, rows_per_group, channels, colsum_ws, (hipStream_t)stream)) return rc;
    return heal_split_attn_weights_from_colsum(colsum_ws, 1, groups, rows_per_group, channels, w_out, b_out, fc1, ln_gamma, ln_beta,
                                               eps, fc2, scale, bias, stream);
}
