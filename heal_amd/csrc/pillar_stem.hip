// LiDAR pillar stem (round 6): the first BasicBlock convolutions of the PointPillars ResNetBEVBackbone read the PILLARS, not a canvas.
//
// Reference: PointPillarScatter.forward (opencood/models/sub_modules/point_pillar_scatter.py:19-76) writes the [n, 64, ny, nx] canvas
// (67 MB per agent, 96 % zeros: ~11 k pillars in 262 144 cells), and ResNetBEVBackbone's first block (base_bev_backbone_resnet.py:88-109,
// resblock.py:18-64: conv1 = 3x3 stride 2 pad 1 + BN + ReLU, downsample = 1x1 stride 2 + BN) reads it back.  Round 5 ran that as
// k_canvas (201 MB written for three agents, ~40 us) + heal_conv3x3 stride 2 (175 us, 14.5 GFLOP of which 0.7 meet a non-zero input)
// + heal_conv1x1 stride 2 (45 us).  Here ONE launch computes both convolutions from the pillar feature rows [M, 64] through the
// cell -> pillar map that K2's PFN stage builds anyway (heal_pfn_pillars); the canvas is never materialised.
//
//   block  = an 8 x 8 tile of output pixels (6.4 m square) x all 64 + 64 output channels; 256 threads = 4 waves.
//   sparsity at TILE x TAP granularity: a tap none of the tile's 64 pixels has (its 8 x 8 stride-2 cell set is empty) is skipped, a
//            tile with no pillar under its 17 x 17 cell footprint writes relu(bias) / bias and leaves.  On the synthetic 64-line
//            sweeps 48 % of the tiles are live and 41 % of the dense tap iterations remain (row strips of 64 pixels: 84 % / 59 %;
//            4 x 4 tiles: 29 % / 22 % but a quarter of the MFMA tile).
//   GEMM   = implicit, M = 64 pixels, N = 64 conv1 channels (+ 64 downsample channels as a tenth "tap" at the centre), K = 64 per
//            tap, on v_mfma_f32_32x32x2_f32: wave (wm, wn) owns pixels [32 wm, +32) x channels [32 wn, +32), both operands staged
//            [row][k] with a 68-float stride (conflict-free 16-B fragment reads), double-buffered, the next tap's pillar rows and
//            weight tile in flight during the MFMAs.  A pixel's operand row at a tap is ONE contiguous 256-B pillar row (or zeros).
//   order  = taps ascending, channels ascending inside a tap: a fixed summation order, bit-reproducible; an absent tap adds nothing
//            (the dense convolution adds exact zeros there), so the result differs from the dense kernels by rounding order only.
// Weights pre-laid by the host exactly as for heal_bev_stem_block (ops.stem_fragments): main [9][1][64][64] (tap, cout, k),
// downsample [1][64][64]; BatchNorm folded into weight / bias by the caller.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PS_C = 64;                 // pillar feature channels = K per tap
constexpr int PS_RS = PS_C + 4;          // LDS row stride (floats)
constexpr int PS_KH = PS_C / 2;          // k-steps per tap: lane half h multiplies k = PS_KH h + s
constexpr int PS_T = 8;                  // output tile edge

__global__ __launch_bounds__(256, 2) void k_pillar_stem(const float* __restrict__ pillars, const int* __restrict__ cell_map,
                                                       int nx, int ny, const float* __restrict__ wmain,
                                                       const float* __restrict__ bmain, const float* __restrict__ wds,
                                                       const float* __restrict__ bds, float* __restrict__ out_main,
                                                       float* __restrict__ out_id, int Ho, int Wo, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float s_all[4 * 64 * PS_RS];
    __shared__ int s_id[9][64];
    __shared__ int s_tapmask;
    float (*sA)[64 * PS_RS] = reinterpret_cast<float (*)[64 * PS_RS]>(s_all);
    float (*sB)[64 * PS_RS] = reinterpret_cast<float (*)[64 * PS_RS]>(s_all + 2 * 64 * PS_RS);
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = l & 31, h = l >> 5;
    int bx_ = blockIdx.x;
    const int tx = bx_ % tiles_x; bx_ /= tiles_x;
    const int ty = bx_ % tiles_y, b = bx_ / tiles_y;
    const int oy0 = ty * PS_T, ox0 = tx * PS_T;
    if (tid == 0) s_tapmask = 0;
    __syncthreads();

    // pillar row of every (tap, pixel) of the tile, or -1: 576 map reads per block (a 17 x 17 cell footprint)
    const int* __restrict__ map_b = cell_map + (size_t)b * ny * nx;
    {
        int mask = 0;
        for (int e = tid; e < 9 * 64; e += 256) {
            const int t = e >> 6, p = e & 63;
            const int oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
            const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
            int id = -1;
            if (oy < Ho && ox < Wo && iy >= 0 && iy < ny && ix >= 0 && ix < nx) id = map_b[(size_t)iy * nx + ix];
            s_id[t][p] = id;
            if (id >= 0) mask |= 1 << t;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o, 64);
        if (l == 0 && mask) atomicOr(&s_tapmask, mask);
    }
    __syncthreads();
    // virtual tap 9 = the centre tap once more, against the DOWNSAMPLE weights
    int tapmask = s_tapmask;
    tapmask |= ((tapmask >> 4) & 1) << 9;
    const int n_it = __popc(tapmask);

    f32x16 acc_m[2], acc_d[2];               // two accumulators per output: consecutive MFMAs never wait for each other
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_m[0][r] = 0.f; acc_m[1][r] = 0.f; acc_d[0][r] = 0.f; acc_d[1][r] = 0.f; }

    // staging role: pixel rows r0 and r0 + 32 of the A tile, 16-B column c4 of each 128-B half row; the weight tile [64][64] as
    // float4 index tid + 256 q.  Staging registers are NAMED values (arrays written in one branch and read in another go to scratch).
    const int r0 = tid >> 3, c4 = tid & 7;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a00 = z4, a01 = z4, a10 = z4, a11 = z4, b0 = z4, b1 = z4, b2 = z4, b3 = z4;
    int on0 = 0, on1 = 0;
    int rem = tapmask, it_tap = 0;
    auto next_tap = [&]() { it_tap = rem ? __builtin_ctz(rem) : 0; rem &= rem - 1; };
    auto load = [&](int t) {
        const int tp = t == 9 ? 4 : t;
        const float* wsrc = t == 9 ? wds : wmain + (size_t)t * 64 * PS_C;
        const int i0 = s_id[tp][r0], i1 = s_id[tp][r0 + 32];
        on0 = i0 >= 0; on1 = i1 >= 0;
        const float* p0 = pillars + (size_t)max(i0, 0) * PS_C + c4 * 4;      // unconditional loads on a clamped row, masked at the
        const float* p1 = pillars + (size_t)max(i1, 0) * PS_C + c4 * 4;      // LDS store (a predicated load drains the queue)
        a00 = *reinterpret_cast<const float4*>(p0);
        a01 = *reinterpret_cast<const float4*>(p0 + 32);
        a10 = *reinterpret_cast<const float4*>(p1);
        a11 = *reinterpret_cast<const float4*>(p1 + 32);
        b0 = *reinterpret_cast<const float4*>(wsrc + tid * 4);
        b1 = *reinterpret_cast<const float4*>(wsrc + (tid + 256) * 4);
        b2 = *reinterpret_cast<const float4*>(wsrc + (tid + 512) * 4);
        b3 = *reinterpret_cast<const float4*>(wsrc + (tid + 768) * 4);
    };
    auto put_a = [&](float* dst, const float4& v, bool on) {   // per component: float4 selects are lowered through scratch
        *reinterpret_cast<float4*>(dst) = make_float4(on ? v.x : 0.f, on ? v.y : 0.f, on ? v.z : 0.f, on ? v.w : 0.f);
    };
    auto put_b = [&](float* sb, int q, const float4& v) {
        const int i4 = tid + 256 * q, row = i4 >> 4, col4 = i4 & 15;
        *reinterpret_cast<float4*>(sb + row * PS_RS + col4 * 4) = v;
    };
    auto store = [&](int buf) {
        put_a(&sA[buf][r0 * PS_RS + c4 * 4], a00, on0);
        put_a(&sA[buf][r0 * PS_RS + 32 + c4 * 4], a01, on0);
        put_a(&sA[buf][(r0 + 32) * PS_RS + c4 * 4], a10, on1);
        put_a(&sA[buf][(r0 + 32) * PS_RS + 32 + c4 * 4], a11, on1);
        put_b(sB[buf], 0, b0);
        put_b(sB[buf], 1, b1);
        put_b(sB[buf], 2, b2);
        put_b(sB[buf], 3, b3);
    };
    if (n_it > 0) {
        next_tap();
        load(it_tap);
        store(0);
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1, t_cur = it_tap;
        next_tap();                                       // it_tap now names iteration it + 1
        if (it + 1 < n_it) load(it_tap);                  // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        float af[PS_KH], bf[PS_KH];
        {
            const float* p = &sA[buf][(wm * 32 + li) * PS_RS + PS_KH * h];
            const float* q = &sB[buf][(wn * 32 + li) * PS_RS + PS_KH * h];
#pragma unroll
            for (int k = 0; k < PS_KH / 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * k);
                af[4 * k] = v.x; af[4 * k + 1] = v.y; af[4 * k + 2] = v.z; af[4 * k + 3] = v.w;
                const float4 u = *reinterpret_cast<const float4*>(q + 4 * k);
                bf[4 * k] = u.x; bf[4 * k + 1] = u.y; bf[4 * k + 2] = u.z; bf[4 * k + 3] = u.w;
            }
        }
        if (t_cur == 9) {
#pragma unroll
            for (int s = 0; s < PS_KH; ++s) acc_d[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc_d[s & 1], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < PS_KH; ++s) acc_m[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc_m[s & 1], 0, 0, 0);
        }
        if (it + 1 < n_it) store(buf ^ 1);
        __syncthreads();
    }

    // epilogue: accumulators (32x32 C/D layout: column = lane & 31 = channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = pixel)
    // -> sC[channel][pixel] (stride 65) -> 16-B NCHW stores (4 pixels of one tile row); conv1 half first, then the downsample half
    constexpr int CS = 65;
    float* sC = s_all;
    static_assert(sizeof(s_all) >= (size_t)64 * CS * 4, "epilogue tile must fit the operand rings");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if (n_it > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                sC[(wn * 32 + li) * CS + px] = half ? acc_d[0][r] + acc_d[1][r] : acc_m[0][r] + acc_m[1][r];
            }
        }
        __syncthreads();
        const float* __restrict__ bias = half ? bds : bmain;
        float* __restrict__ outp = half ? out_id : out_main;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, co = idx >> 4, p4 = idx & 15;
            const int oy = oy0 + (p4 >> 1), ox = ox0 + (p4 & 1) * 4;
            if (oy >= Ho || ox >= Wo) continue;   // Wo % 4 == 0 (host)
            const float bv = bias ? bias[co] : 0.f;
            float4 v = make_float4(bv, bv, bv, bv);
            if (n_it > 0) {
                const float* sp = &sC[co * CS + p4 * 4];
                v = make_float4(sp[0] + bv, sp[1] + bv, sp[2] + bv, sp[3] + bv);
            }
            if (!half) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(outp + (((size_t)b * 64 + co) * Ho + oy) * Wo + ox) = v;
        }
    }
}


// ---- v2: pixel-compacted, barrier-free per wave ----------------------------------------------------------------------------------------
// v1 above skips at TILE x TAP granularity and still multiplies 41 % of the dense tap iterations (141 us for three agents: 6.6 GFLOP of
// MFMA work at 30 % utilisation) although only 12 % of the output pixels see a pillar at all.  v2 compacts the ACTIVE output pixels of a
// tile and multiplies only those, in groups of 16, on the transposed formulation D^T[cout][pixel] = W_tap . X^T of v_mfma_f32_16x16x4_f32:
//   block  = a 2 x 32 tile of output pixels (128-B output runs), 4 waves; wave w owns output channels [16 w, 16 w + 16) of conv1 AND of the
//            downsample, for every pixel of the tile -- no operand is shared between waves through LDS, so after the tile's prologue
//            (map reads, active-pixel compaction by one ballot) the waves never meet again until the output tile is written.
//   A      = the tap's weights for the wave's 16 channels, 16 registers per lane, read from the host-pre-laid fragment array (L2) once per tap;
//   B      = the pillar row of the lane's pixel: lane (lk, ln) takes channels [16 lk, 16 lk + 16) of pixel ln's pillar at this tap as four 16-B
//            loads straight from global memory (the reduction index is permuted: step ks multiplies channel 16 lk + ks on both operands);
//   16 MFMAs per (group, tap), two partial accumulators (even / odd steps) so that consecutive MFMAs are independent; the centre tap feeds
//   the downsample accumulators from the SAME B registers.  ~2 000 groups x ~9 taps x 16 MFMAs for three agents: the matrix work is
//   microseconds; what the kernel must do is write the 100 MB of outputs (every tile, live or not, writes its 64 x 128 values once).
//   epilogue: per wave through a private LDS slice [16 channels][64 pixels]: bias (ReLU on conv1) everywhere, the active pixels' sums
//   scattered in, 16-B NCHW stores.
// Measured (scripts/pillar_stem_bench.py, three 64-line agents): 141 (v1) -> 100 us; anatomy (HEAL_PS_DBG): prologue + epilogue with
// all 100 MB of stores 27 us, the tap loop 73 us -- a chain of dependent gathers per live tile.  Requesting the rows of all four groups
// of a tap before its first MFMA (64 more registers, 3 instead of 5 waves per SIMD) measured 109 us: not kept.
constexpr int PS2_TH = 2, PS2_TW = 32;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void k_pillar_stem2(const float* __restrict__ pillars, const int* __restrict__ cell_map, int nx,
                                                     int ny, const float4* __restrict__ wmain /*[9][4][64][4] float4*/,
                                                     const float* __restrict__ bmain, const float4* __restrict__ wds /*[4][64][4]*/,
                                                     const float* __restrict__ bds, float* __restrict__ out_main,
                                                     float* __restrict__ out_id, int Ho, int Wo, int tiles_x, int tiles_y, int dbg) {
    // dbg (HEAL_PS_DBG, timing anatomy only -- results invalid): 1 skip the tap loop, 2 skip the output stores, 4 skip the map reads,
    // 16 no pillar-row gathers (row 0 for everybody), 32 no weight loads
    constexpr int OS = 68;                   // row stride of the per-wave output slice
    __shared__ int s_id[9][64];
    __shared__ int s_list[64];               // compacted active pixels
    __shared__ int s_bits[64];               // 9-bit tap mask per pixel
    __shared__ int s_nact;
    __shared__ __attribute__((aligned(16))) float s_out[4][16 * OS];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lk = l >> 4, ln = l & 15;
    int bx_ = blockIdx.x;
    const int tx = bx_ % tiles_x; bx_ /= tiles_x;
    const int ty = bx_ % tiles_y, b = bx_ / tiles_y;
    const int oy0 = ty * PS2_TH, ox0 = tx * PS2_TW;
    const int* __restrict__ map_b = cell_map + (size_t)b * ny * nx;
    for (int e = tid; e < 9 * 64; e += 256) {
        const int t = e >> 6, p = e & 63;
        const int oy = oy0 + p / PS2_TW, ox = ox0 + p % PS2_TW;
        const int iy = 2 * oy + t / 3 - 1, ix = 2 * ox + t % 3 - 1;
        int id = -1;
        if (!(dbg & 4) && oy < Ho && ox < Wo && iy >= 0 && iy < ny && ix >= 0 && ix < nx) id = map_b[(size_t)iy * nx + ix];
        s_id[t][p] = id;
    }
    __syncthreads();
    if (w == 0) {                            // one wave = the tile's 64 pixels: tap bits, compaction by ballot
        int bits = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) bits |= (int)(s_id[t][l] >= 0) << t;
        s_bits[l] = bits;
        const unsigned long long act = __ballot(bits != 0);
        if (bits) s_list[__popcll(act & lanemask_lt())] = l;
        if (l == 0) s_nact = __popcll(act);
    }
    __syncthreads();
    const int n_act = s_nact;
    const int n_groups = (n_act + 15) >> 4;  // <= 4

    f32x4 am[4][2], ad[4][2];                // [group][even / odd k-steps]
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 2; ++q) { am[g][q] = f32x4{0.f, 0.f, 0.f, 0.f}; ad[g][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    int px[4], gbits[4];                     // the lane's pixel in group g (-1: none), the group's tap mask (wave-uniform)
    int tile_bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bool on = 16 * g + ln < n_act;
        px[g] = on ? s_list[min(16 * g + ln, 63)] : -1;
        int bt = on ? s_bits[max(px[g], 0)] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) bt |= __shfl_xor(bt, o, 64);      // OR over the 16 pixels (lanes of equal lk)
        gbits[g] = __builtin_amdgcn_readfirstlane(bt);
        tile_bits |= gbits[g];
    }
    if (dbg & 1) tile_bits = 0;
    for (int t = 0; t < 9; ++t) {
        if (!((tile_bits >> t) & 1)) continue;                              // block-uniform
        float4 a4[4], d4[4];
        const float4* wa = wmain + ((size_t)(((dbg & 32) ? 0 : t) * 4 + w) * 64 + l) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] = wa[q];
        if (t == 4) {
            const float4* wd_ = wds + ((size_t)w * 64 + l) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = wd_[q];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!((gbits[g] >> t) & 1)) continue;                           // wave-uniform
            const int id = px[g] >= 0 ? s_id[t][px[g]] : -1;
            const float4* row = reinterpret_cast<const float4*>(pillars + (size_t)((dbg & 16) ? 0 : max(id, 0)) * PS_C + 16 * lk);
            float4 b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = row[q];                     // unconditional on a clamped row, masked below
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                   // (per component: a float4 select goes through scratch)
                b4[q].x = id < 0 ? 0.f : b4[q].x; b4[q].y = id < 0 ? 0.f : b4[q].y;
                b4[q].z = id < 0 ? 0.f : b4[q].z; b4[q].w = id < 0 ? 0.f : b4[q].w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                am[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].x, b4[q].x, am[g][0], 0, 0, 0);
                am[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].y, b4[q].y, am[g][1], 0, 0, 0);
                am[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].z, b4[q].z, am[g][0], 0, 0, 0);
                am[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].w, b4[q].w, am[g][1], 0, 0, 0);
            }
            if (t == 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ad[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[q].x, b4[q].x, ad[g][0], 0, 0, 0);
                    ad[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[q].y, b4[q].y, ad[g][1], 0, 0, 0);
                    ad[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[q].z, b4[q].z, ad[g][0], 0, 0, 0);
                    ad[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[q].w, b4[q].w, ad[g][1], 0, 0, 0);
                }
            }
        }
    }

    // epilogue (per wave, its own LDS slice): D[row = 4 lk + r][col = ln] = channel 16 w + 4 lk + r of pixel px[g]
    float* so = s_out[w];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float* __restrict__ bias = half ? bds : bmain;
        float* __restrict__ outp = half ? out_id : out_main;
        __syncthreads();                     // (block-uniform control flow: every wave passes the same barriers)
#pragma unroll
        for (int i = 0; i < 4; ++i) {        // background: bias (ReLU on conv1) in every pixel of the slice
            const int idx = l + 64 * i, co = idx >> 4, p4 = idx & 15;
            float bv = bias ? bias[16 * w + co] : 0.f;
            if (!half) bv = fmaxf(bv, 0.f);
            *reinterpret_cast<float4*>(&so[co * OS + p4 * 4]) = make_float4(bv, bv, bv, bv);
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g >= n_groups || px[g] < 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * lk + r;
                const float bv = bias ? bias[16 * w + co] : 0.f;
                float v = (half ? ad[g][0][r] + ad[g][1][r] : am[g][0][r] + am[g][1][r]) + bv;
                if (!half) v = fmaxf(v, 0.f);
                so[co * OS + px[g]] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = l + 64 * i, co = idx >> 4, p4 = idx & 15;
            const int oy = oy0 + (p4 * 4) / PS2_TW, ox = ox0 + (p4 * 4) % PS2_TW;
            if (oy >= Ho || ox >= Wo || (dbg & 2)) continue;   // Wo % 4 == 0 (host)
            *reinterpret_cast<float4*>(outp + (((size_t)b * 64 + 16 * w + co) * Ho + oy) * Wo + ox) =
                *reinterpret_cast<const float4*>(&so[co * OS + p4 * 4]);
        }
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_pillar_stem_block(const float* pillar_feat, const int32_t* cell_map, int n_agents, int channels, int ny,
                                      int nx, const float* w_main, const float* b_main, const float* w_down,
                                      const float* b_down, int weight_layout, float* out_main, float* out_identity, void* stream) {
    HEAL_REQUIRE(channels == PS_C, "pillar_stem_block: pillar features must have 64 channels (got %d)", channels);
    HEAL_REQUIRE(n_agents >= 1 && ny >= 1 && nx >= 1, "pillar_stem_block: bad grid");
    HEAL_REQUIRE(pillar_feat && cell_map && w_main && w_down && out_main && out_identity, "pillar_stem_block: null pointer");
    HEAL_REQUIRE(weight_layout == 0 || weight_layout == 1, "pillar_stem_block: weight_layout 0 (lane fragments) | 1 (tap-major tiles)");
    const int Ho = (ny - 1) / 2 + 1, Wo = (nx - 1) / 2 + 1;
    HEAL_REQUIRE(Wo % 4 == 0, "pillar_stem_block: output width must be a multiple of 4 (got %d)", Wo);
    HEAL_REQUIRE((((uintptr_t)pillar_feat | (uintptr_t)w_main | (uintptr_t)w_down | (uintptr_t)out_main | (uintptr_t)out_identity) & 15) == 0,
                 "pillar_stem_block: 16-B alignment");
    if (weight_layout == 1) {                // v1: 8 x 8 tiles, tile x tap skipping, 32x32x2 MFMA through LDS (kept for A/B)
        const int tiles_x = ceil_div(Wo, PS_T), tiles_y = ceil_div(Ho, PS_T);
        const int64_t blocks = (int64_t)n_agents * tiles_y * tiles_x;
        HEAL_REQUIRE(blocks < (1ll << 31), "pillar_stem_block: grid too large");
        HEAL_LAUNCH_EV(k_pillar_stem, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pillar_feat, cell_map, nx, ny, w_main,
                       b_main, w_down, b_down, out_main, out_identity, Ho, Wo, tiles_x, tiles_y);
    } else {
        const int tiles_x = ceil_div(Wo, PS2_TW), tiles_y = ceil_div(Ho, PS2_TH);
        const int64_t blocks = (int64_t)n_agents * tiles_y * tiles_x;
        HEAL_REQUIRE(blocks < (1ll << 31), "pillar_stem_block: grid too large");
        HEAL_LAUNCH_EV(k_pillar_stem2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pillar_feat, cell_map, nx, ny,
                       reinterpret_cast<const float4*>(w_main), b_main, reinterpret_cast<const float4*>(w_down), b_down, out_main,
                       out_identity, Ho, Wo, tiles_x, tiles_y, HEAL_DEBUG_ENV("HEAL_PS_DBG"));
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}
