// Pointwise convolution on the 128 x 128 x 32 fp32 core (v_mfma_f32_32x32x2_f32) -- the MFMA-bound shapes of heal_conv1x1.
//
// Same operation and call sites as conv1x1.hip (ResNeXt conv1 / conv3 of the pyramid stages, resblock.py:95-121; the deblocks
// ConvTranspose2d(kernel = stride) as a pointwise convolution + depth-to-space, base_bev_backbone_resnet.py:49-74; ConvNeXt
// pwconv1 / pwconv2, feature_alignnet_modules.py:299-344):  Y[Cout, HW] = act(W[Cout, Cin] X[Cin, HW] + b (+ residual)), NCHW.
//
// Why a second kernel: the 64 x 64 tiles of k_conv1x1 (16x16x4 MFMA, A fragments from L2, 8 waves / SIMD) win where the
// convolution is HBM- or latency-bound, but on the MFMA-bound shapes (Cin, Cout >= 128: 62 % of the family's time) they sit at
// 0.55-0.59 of the fp32 matrix peak with SQ_VALU_MFMA_BUSY 0.48: a 64 x 64 x 32 block step moves 16 KB through LDS / L2 for
// 262 kFLOP, and every 16x16x4 MFMA (32 cycles) needs its own B fragment read.  heal_linear's core holds 0.64-0.72 on the same
// matrix sizes; this is that core with the operands of a convolution:
//   * block = 4 waves = BM output channels x 128 pixels (BM = 128: waves 2 x 2 of 64 x 64, four 32x32 accumulators each;
//     BM = 64: waves 2 x 2 of 32 x 64), K chunks of 32 through double-buffered LDS, next chunk's global loads in flight under the
//     MFMAs (one barrier per chunk);
//   * A = weights [Cout][Cin] row-major as nn.Conv2d stores them (no fragment pre-layout): staged [row][k], row stride 36,
//     fragments by ds_read_b128 (lane half h multiplies k = 16 h + s at step s -- the reduction order both operands share);
//   * B = activations [Cin][HW]: a chunk is 32 rows of 512 contiguous bytes, stored as they come ([k][pixel], 16-B stores) and
//     read as the MFMA wants them: lane (pixel = l & 31, h) reads sB[16 h + s][pixel] -- 32 consecutive words per half-wave,
//     conflict-free ds_read_b32, one per (k-step, 32-pixel tile);
//   * epilogue through LDS (the operand rings are free by then): accumulators -> sC[channel][pixel] -> every thread owns 16-B
//     pieces of output rows: bias, residual, ReLU | SiLU | GELU, and the depth-to-space + channel-offset addressing of the
//     deblocks, as 16-B accesses.
// Roofline: fp32 MFMA (157.3 TFLOP/s).  Dispatch (ops.conv1x1): Cin % 32 == 0, Cout % 64 == 0, stride 1, no input gate, >= 256
// blocks; everything else stays on k_conv1x1.
#include <stdlib.h>
#include "../common.h"
#include "../../../include/heal_amd.h"
#include "../../../include/heal_amd_experimental.h"

namespace heal {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int C1T_BN = 128, C1T_BK = 32, C1T_RS = 36, C1T_LDB = 128;

template <int BM>
__global__ __launch_bounds__(256, 2) void k_conv1x1_t(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ res, int Cin,
                                                     int Cout, int HW, int Wo, int act, int d2s_k, int d2s_ctot, int d2s_coff,
                                                     float* __restrict__ y) {
    constexpr int MT = BM / 64;                        // 32-row m-tiles per wave
    constexpr int A_F4 = BM * C1T_BK / 4 / 256;        // float4 of the weight tile per thread (4 | 2)
    constexpr int SA = BM * C1T_RS, SB = C1T_BK * C1T_LDB;
    constexpr int CS = C1T_BN + 4;                     // epilogue tile row stride
    constexpr int S_ALL = (2 * SA + 2 * SB) > BM * CS ? (2 * SA + 2 * SB) : BM * CS;
    __shared__ __attribute__((aligned(16))) float s_all[S_ALL];
    float* sA = s_all;                 // [2][BM][36]
    float* sB = s_all + 2 * SA;        // [2][32][128]
    const Block3 bk = xcd_block();     // x: Cout block (fastest: blocks sharing a pixel tile are neighbours), y: pixel tile, z: image
    const int m0 = bk.x * BM, p0 = bk.y * C1T_BN, n = bk.z;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = l & 31, h = l >> 5;
    const float* __restrict__ xin = x + (size_t)n * Cin * HW;

    // staging roles.  A: float4 index i = tid + 256 q -> row i / 8 (channel), 16-B column i % 8.  B: i -> row i / 32 (k), column i % 32.
    const float* a_src[A_F4];
    int a_dst[A_F4];
#pragma unroll
    for (int q = 0; q < A_F4; ++q) {
        const int i = tid + 256 * q, row = i >> 3, c4 = i & 7;
        a_src[q] = w + (size_t)min(m0 + row, Cout - 1) * Cin + c4 * 4;
        a_dst[q] = row * C1T_RS + c4 * 4;
    }
    const int b_c4 = tid & 31, b_row = tid >> 5;         // rows b_row + 8 q, q < 4
    const int px = min(p0 + b_c4 * 4, HW - 4);           // HW % 4 == 0 (host): clamped, tail pixels are never stored
    const float* b_src = xin + (size_t)b_row * HW + px;

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][nn][r] = 0.f;

    // staging registers as NAMED float4 values: arrays of float4 that are written in one branch and read in another end up in
    // scratch memory (8 scratch stores + 8 loads per chunk in the first version of this kernel)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define C1T_LOAD(k0_)                                                                      \
    ra0 = *reinterpret_cast<const float4*>(a_src[0] + (k0_));                              \
    ra1 = *reinterpret_cast<const float4*>(a_src[1] + (k0_));                              \
    if constexpr (A_F4 == 4) {                                                             \
        ra2 = *reinterpret_cast<const float4*>(a_src[2] + (k0_));                          \
        ra3 = *reinterpret_cast<const float4*>(a_src[3] + (k0_));                          \
    }                                                                                      \
    rb0 = *reinterpret_cast<const float4*>(b_src + (size_t)((k0_) + 0) * HW);              \
    rb1 = *reinterpret_cast<const float4*>(b_src + (size_t)((k0_) + 8) * HW);              \
    rb2 = *reinterpret_cast<const float4*>(b_src + (size_t)((k0_) + 16) * HW);             \
    rb3 = *reinterpret_cast<const float4*>(b_src + (size_t)((k0_) + 24) * HW);
#define C1T_STORE(buf_)                                                                    \
    *reinterpret_cast<float4*>(&sA[(buf_) * SA + a_dst[0]]) = ra0;                         \
    *reinterpret_cast<float4*>(&sA[(buf_) * SA + a_dst[1]]) = ra1;                         \
    if constexpr (A_F4 == 4) {                                                             \
        *reinterpret_cast<float4*>(&sA[(buf_) * SA + a_dst[2]]) = ra2;                     \
        *reinterpret_cast<float4*>(&sA[(buf_) * SA + a_dst[3]]) = ra3;                     \
    }                                                                                      \
    *reinterpret_cast<float4*>(&sB[(buf_) * SB + (b_row + 0) * C1T_LDB + b_c4 * 4]) = rb0; \
    *reinterpret_cast<float4*>(&sB[(buf_) * SB + (b_row + 8) * C1T_LDB + b_c4 * 4]) = rb1; \
    *reinterpret_cast<float4*>(&sB[(buf_) * SB + (b_row + 16) * C1T_LDB + b_c4 * 4]) = rb2; \
    *reinterpret_cast<float4*>(&sB[(buf_) * SB + (b_row + 24) * C1T_LDB + b_c4 * 4]) = rb3;

    const int n_chunks = Cin / C1T_BK;
    C1T_LOAD(0)
    C1T_STORE(0)
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) { C1T_LOAD((c + 1) * C1T_BK) }   // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        float af[MT][16];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* p = &sA[buf * SA + (wm * (BM / 2) + m * 32 + li) * C1T_RS + 16 * h];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                af[m][4 * q] = v.x; af[m][4 * q + 1] = v.y; af[m][4 * q + 2] = v.z; af[m][4 * q + 3] = v.w;
            }
        }
        const float* bp = &sB[buf * SB + (16 * h) * C1T_LDB + wn * 64 + li];
        float bf[2][2];                       // B fragments of k-step s + 1 are read while the MFMAs of k-step s run
        bf[0][0] = bp[0];
        bf[0][1] = bp[32];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) {
                bf[(s + 1) & 1][0] = bp[(s + 1) * C1T_LDB];
                bf[(s + 1) & 1][1] = bp[(s + 1) * C1T_LDB + 32];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][s], bf[s & 1][nn], acc[m][nn], 0, 0, 0);
        }
        if (c + 1 < n_chunks) { C1T_STORE(buf ^ 1) }
        __syncthreads();
    }
#undef C1T_LOAD
#undef C1T_STORE

    // ---- epilogue: accumulators (C/D layout of 32x32: column = lane & 31 = pixel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) =
    // channel) -> sC[channel][pixel] -> 16-B pieces of output rows
    float* sC = s_all;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * (BM / 2) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CS + wn * 64 + nn * 32 + li] = acc[m][nn][r];
    __syncthreads();
    const int c4 = tid & 31, p = p0 + c4 * 4;
    if (p >= HW) return;
    float* __restrict__ yout = y + (size_t)n * Cout * HW;
    const float* __restrict__ rin = res ? res + (size_t)n * Cout * HW : nullptr;
#pragma unroll 4
    for (int i = 0; i < BM / 8; ++i) {
        const int row = (tid >> 5) + 8 * i, co = m0 + row;
        if (co >= Cout) break;
        float4 v = *reinterpret_cast<const float4*>(&sC[row * CS + c4 * 4]);
        const float bv = bias ? bias[co] : 0.f;
        const size_t o = (size_t)co * HW + p;
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        if (rin) {
            const float4 q = *reinterpret_cast<const float4*>(rin + o);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (act == 1) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (act == 2) {
            v.x = v.x / (1.f + expf(-v.x)); v.y = v.y / (1.f + expf(-v.y));
            v.z = v.z / (1.f + expf(-v.z)); v.w = v.w / (1.f + expf(-v.w));
        } else if (act == 3) {  // exact GELU: 0.5 x (1 + erf(x / sqrt(2)))
            v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752f));
            v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752f));
            v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752f));
            v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752f));
        }
        if (d2s_k == 0) {
            *reinterpret_cast<float4*>(yout + o) = v;
        } else {   // depth-to-space INTO A SLICE of a wider NCHW tensor (conv1x1.hip, out_pm = 2)
            const int kk = d2s_k * d2s_k, Ho = HW / Wo;
            const int cc = co / kk, r = co - cc * kk, dy = r / d2s_k, dx = r - dy * d2s_k;
            const int hh = p / Wo, ww = p - hh * Wo;      // Wo % 4 == 0 (host): the four pixels share a row
            float* dst = y + (((size_t)n * d2s_ctot + d2s_coff + cc) * ((size_t)Ho * d2s_k) + (size_t)hh * d2s_k + dy) *
                                 ((size_t)Wo * d2s_k) + (size_t)ww * d2s_k + dx;
            if (d2s_k == 1) {
                *reinterpret_cast<float4*>(dst) = v;
            } else {
                dst[0] = v.x; dst[d2s_k] = v.y; dst[2 * d2s_k] = v.z; dst[3 * d2s_k] = v.w;
            }
        }
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_conv1x1_tiled_supported(int n, int cin, int cout, int H, int W) {
    const long long hw = (long long)H * W;
    if (n < 1 || cin < 32 || cin % 32 != 0 || cout % 64 != 0 || hw % 4 != 0 || hw < 128) return 0;
    const int bm = cout % 128 == 0 ? 128 : 64;
    const long long blocks = (long long)(cout / bm) * ((hw + C1T_BN - 1) / C1T_BN) * n;
    return blocks >= 256 ? 1 : 0;
}

extern "C" int heal_conv1x1_tiled(const float* x, const float* weight, const float* bias, const float* residual, int n, int cin,
                                  int cout, int H, int W, int act, int d2s_k, int dst_channels, int dst_channel_offset, float* y,
                                  void* stream) {
    HEAL_REQUIRE(x && weight && y, "conv1x1_tiled: null pointer");
    HEAL_REQUIRE(n >= 1 && n <= 65535 && cin >= 32 && cin % 32 == 0 && cout >= 64 && cout % 64 == 0,
                 "conv1x1_tiled: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d -> %d)", cin, cout);
    HEAL_REQUIRE(((long long)H * W) % 4 == 0 && (long long)H * W >= 4, "conv1x1_tiled: H*W must be a multiple of 4");
    HEAL_REQUIRE(act >= 0 && act <= 3, "conv1x1_tiled: act must be 0 (none), 1 (ReLU), 2 (SiLU) or 3 (GELU)");
    HEAL_REQUIRE((((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)residual) & 15) == 0, "conv1x1_tiled: 16-B alignment");
    if (d2s_k > 0) {
        HEAL_REQUIRE(d2s_k <= 8 && cout % (d2s_k * d2s_k) == 0 && W % 4 == 0 && residual == nullptr,
                     "conv1x1_tiled: depth-to-space needs Cout %% k^2 == 0, W %% 4 == 0 and no residual");
        HEAL_REQUIRE(dst_channel_offset >= 0 && dst_channel_offset + cout / (d2s_k * d2s_k) <= dst_channels,
                     "conv1x1_tiled: channel slice outside the destination");
    }
    const int HW = H * W;
    hipStream_t s = (hipStream_t)stream;
    if (cout % 128 == 0)
        k_conv1x1_t<128><<<dim3(cout / 128, ceil_div(HW, C1T_BN), n), 256, 0, s>>>(x, weight, bias, residual, cin, cout, HW, W, act,
                                                                                    d2s_k, dst_channels, dst_channel_offset, y);
    else
        k_conv1x1_t<64><<<dim3(cout / 64, ceil_div(HW, C1T_BN), n), 256, 0, s>>>(x, weight, bias, residual, cin, cout, HW, W, act,
                                                                                  d2s_k, dst_channels, dst_channel_offset, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}
