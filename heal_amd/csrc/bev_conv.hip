// K7 helpers for the dense BEV stacks -- the parts of the ResNeXt pyramid where a vendor convolution is
// far from the roofline:
//
//  * heal_grouped_conv3x3: the 32-group 3x3 convolutions of the ResNeXt bottlenecks
//    (opencood/models/sub_modules/resblock.py:90-98 via pyramid_fuse.py:71-79: groups 32, 4/8/16 channels
//    per group) with the folded BatchNorm bias and ReLU as epilogue.  With 4..16 channels per group there
//    is no GEMM worth a matrix core: it is an HBM-bound stencil (read the map once, write it once).  A
//    block owns a 32x8 output tile of one (image, group): the input patch (with halo) goes through LDS,
//    every thread keeps all output channels of its pixel in registers, weights are block-uniform.
//  * heal_bias_act: y = act(x + bias[c] (+ residual)) in one pass, in place -- replaces the separate
//    bias-add / residual-add / ReLU kernels that follow a library convolution.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

template <int CG, int STRIDE>
__global__ __launch_bounds__(256) void k_grouped_conv3x3(const float* __restrict__ x,
                                                        const float* __restrict__ w /*[C][CG][3][3]*/,
                                                        const float* __restrict__ bias /*[C] or null*/,
                                                        int C, int H, int W, int Ho, int Wo, int relu,
                                                        float* __restrict__ y) {
    constexpr int TW = 32, TH = 8;
    constexpr int IW = (TW - 1) * STRIDE + 3, IH = (TH - 1) * STRIDE + 3;
    constexpr int CCH = 4;  // input channels staged per pass (keeps LDS small -> several blocks per CU)
    __shared__ float tile[CCH][IH][IW + 1];
    const int G = C / CG;
    const Block3 bk = xcd_block();  // neighbouring tiles (shared halo rows / 128-B lines) on one XCD's L2
    const int n = bk.z / G, g = bk.z - n * G;
    const int ox0 = bk.x * TW, oy0 = bk.y * TH;
    const int ix0 = ox0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1;
    const float* xin = x + ((size_t)n * C + (size_t)g * CG) * H * W;
    // weights of this group: block-uniform addresses -> scalar loads, operands come from SGPRs
    const float* __restrict__ wg = w + (size_t)g * CG * CG * 9;
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
    const int ox = ox0 + tx, oy = oy0 + ty;
    float acc[CG];
#pragma unroll
    for (int co = 0; co < CG; ++co) acc[co] = bias ? bias[g * CG + co] : 0.f;
    for (int c0 = 0; c0 < CG; c0 += CCH) {
        __syncthreads();  // previous pass finished reading the tile
        for (int e = threadIdx.x; e < CCH * IH * IW; e += 256) {
            const int c = e / (IH * IW), r = (e / IW) % IH, col = e % IW;
            const int iy = iy0 + r, ix = ix0 + col;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xin[((size_t)(c0 + c) * H + iy) * W + ix];
            tile[c][r][col] = v;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CCH; ++cc) {
            float v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = tile[cc][ty * STRIDE + k / 3][tx * STRIDE + k % 3];
#pragma unroll
            for (int co = 0; co < CG; ++co) {
                const float* wk = wg + (co * CG + c0 + cc) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[co] = fmaf(v[k], wk[k], acc[co]);
            }
        }
    }
    if (ox < Wo && oy < Ho) {
        float* yo = y + ((size_t)n * C + (size_t)g * CG) * Ho * Wo + (size_t)oy * Wo + ox;
#pragma unroll
        for (int co = 0; co < CG; ++co) {
            const float r = relu ? fmaxf(acc[co], 0.f) : acc[co];
            yo[(size_t)co * Ho * Wo] = r;
        }
    }
}

// Stride-1 variant with packed fp32 math.  The one-pixel kernel above is VALU-bound (measured 33-41 TFLOP/s of the
// 78 TFLOP/s a CU array issues with one v_fma_f32 per lane-cycle): here a thread owns TWO adjacent x pixels and every
// multiply-add is a v_pk_fma_f32 on the (pixel0, pixel1) pair with the block-uniform weight broadcast from an SGPR --
// half the FMA instructions and 12 instead of 18 LDS reads per input channel for the two pixels.
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CG>
__global__ __launch_bounds__(256) void k_grouped_conv3x3_pk(const float* __restrict__ x,
                                                           const float* __restrict__ w /*[C][CG][3][3]*/,
                                                           const float* __restrict__ bias /*[C] or null*/,
                                                           int C, int H, int W, int relu, float* __restrict__ y) {
    constexpr int TW = 64, TH = 8;
    constexpr int IW = TW + 2, IH = TH + 2;
    constexpr int CCH = 4;
    __shared__ float tile[CCH][IH][IW + 1];
    const int G = C / CG;
    const Block3 bk = xcd_block();
    const int n = bk.z / G, g = bk.z - n * G;
    const int ox0 = bk.x * TW, oy0 = bk.y * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;
    const float* xin = x + ((size_t)n * C + (size_t)g * CG) * H * W;
    const float* __restrict__ wg = w + (size_t)g * CG * CG * 9;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ox = ox0 + 2 * tx, oy = oy0 + ty;
    v2f acc[CG];
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        const float b = bias ? bias[g * CG + co] : 0.f;
        acc[co] = v2f{b, b};
    }
    for (int c0 = 0; c0 < CG; c0 += CCH) {
        __syncthreads();
        for (int e = threadIdx.x; e < CCH * IH * IW; e += 256) {
            const int c = e / (IH * IW), r = (e / IW) % IH, col = e % IW;
            const int iy = iy0 + r, ix = ix0 + col;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xin[((size_t)(c0 + c) * H + iy) * W + ix];
            tile[c][r][col] = v;
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CCH; ++cc) {
            v2f a[9];  // (pixel0, pixel1) operands of the 9 taps
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* row = &tile[cc][ty + r][2 * tx];
                const float q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
                a[r * 3 + 0] = v2f{q0, q1};
                a[r * 3 + 1] = v2f{q1, q2};
                a[r * 3 + 2] = v2f{q2, q3};
            }
#pragma unroll
            for (int co = 0; co < CG; ++co) {
                const float* wk = wg + (co * CG + c0 + cc) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float wv = wk[k];
                    acc[co] = __builtin_elementwise_fma(a[k], v2f{wv, wv}, acc[co]);
                }
            }
        }
    }
    if (oy >= H || ox >= W) return;
    float* yo = y + ((size_t)n * C + (size_t)g * CG) * H * W + (size_t)oy * W + ox;
    const bool two = ox + 1 < W;
    const bool vec = two && (W % 2 == 0);
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        v2f r = acc[co];
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); }
        float* d = yo + (size_t)co * H * W;
        if (vec) {
            *reinterpret_cast<float2*>(d) = make_float2(r.x, r.y);
        } else {
            d[0] = r.x;
            if (two) d[1] = r.y;
        }
    }
}

__global__ __launch_bounds__(256) void k_bias_act(float4* __restrict__ x, const float* __restrict__ bias,
                                                 const float4* __restrict__ res, int C, int HW4, long long total4,
                                                 int relu) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < total4; i += stride) {
        const int c = (int)((i / HW4) % C);
        const float b = bias ? bias[c] : 0.f;
        float4 v = x[i];
        v.x += b; v.y += b; v.z += b; v.w += b;
        if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        x[i] = v;
    }
}

// x2 bilinear up-sampling with align_corners=True (nn.Upsample in the Lift-Splat `Up` block,
// lss_submodule.py:21-22), same operation order as ATen's upsample_bilinear2d: src = dst*(in-1)/(out-1),
// out = h0*(w0*a + w1*b) + h1*(w0*c + w1*d).
__global__ __launch_bounds__(256) void k_upsample2x(const float* __restrict__ x, int H, int W, long long total,
                                                   float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int Wo = 2 * W, Ho = 2 * H;
    const int ox = (int)(i % Wo);
    const int oy = (int)((i / Wo) % Ho);
    const long long nc = i / ((long long)Wo * Ho);
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sh * (float)oy, fx = sw * (float)ox;
    const int y1 = (int)fy, x1 = (int)fx;
    const int yp = y1 < H - 1 ? 1 : 0, xp = x1 < W - 1 ? 1 : 0;
    const float ly = fy - (float)y1, lx = fx - (float)x1;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* p = x + nc * (long long)H * W + (long long)y1 * W + x1;
    y[i] = hy * (hx * p[0] + lx * p[xp]) + ly * (hx * p[yp * W] + lx * p[yp * W + xp]);
}

// Depthwise k x k convolution (k = 3 | 5, stride 1 | 2) with explicit top/left padding (TF-style "same"
// padding is asymmetric), folded-BN bias and optional SiLU: the MBConv depthwise stage of the
// EfficientNet-b0 camera trunk (lss_submodule.py:93-105 via efficientnet_pytorch's MBConvBlock).  The
// vendor library falls back to a naive reference kernel for these shapes.
template <int K, int STRIDE>
__global__ __launch_bounds__(256) void k_depthwise(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, int C, int H, int W, int Ho,
                                                  int Wo, int pad_t, int pad_l, int act, float* __restrict__ y,
                                                  float* __restrict__ sums) {
    constexpr int TW = 32, TH = 8;
    constexpr int IW = (TW - 1) * STRIDE + K, IH = (TH - 1) * STRIDE + K;
    __shared__ float tile[IH][IW + 1];
    __shared__ float part[4];
    const Block3 bk = xcd_block();
    const int nc = bk.z;          // n * C + c
    const int c = nc % C;
    const int ox0 = bk.x * TW, oy0 = bk.y * TH;
    const int ix0 = ox0 * STRIDE - pad_l, iy0 = oy0 * STRIDE - pad_t;
    const float* xin = x + (size_t)nc * H * W;
    for (int e = threadIdx.x; e < IH * IW; e += 256) {
        const int r = e / IW, col = e - r * IW;
        const int iy = iy0 + r, ix = ix0 + col;
        tile[r][col] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xin[(size_t)iy * W + ix] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
    const int ox = ox0 + tx, oy = oy0 + ty;
    const bool live = ox < Wo && oy < Ho;
    if (!live && !sums) return;
    const float* __restrict__ wk = w + (size_t)c * K * K;  // block-uniform -> scalar loads
    float acc = bias ? bias[c] : 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc = fmaf(tile[ty * STRIDE + ky][tx * STRIDE + kx], wk[ky * K + kx], acc);
    if (act == 1) acc = fmaxf(acc, 0.f);
    else if (act == 2) acc = acc / (1.f + expf(-acc));  // SiLU
    if (live) y[(size_t)nc * Ho * Wo + (size_t)oy * Wo + ox] = acc;
    if (sums) {
        // the squeeze of the squeeze-excite stage that follows (MBConv: x.mean((2, 3))) rides along: the block sum of the
        // activated outputs goes to sums[n*C + c][tile] -- a plain store per block (atomics into the few (n, c) words
        // serialise at L2: 190 per word in the first MBConv stage tripled that launch's time); k_se_gate adds the tiles up in a
        // fixed order, so the gate is bit-reproducible
        const float ws = wave_sum(live ? acc : 0.f);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ws;
        __syncthreads();
        if (threadIdx.x == 0)
            sums[(size_t)nc * (gridDim.x * gridDim.y) + bk.y * gridDim.x + bk.x] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_depthwise_conv(const float* x, const float* weight, const float* bias, int n, int channels,
                                   int H, int W, int ksize, int stride, int pad_t, int pad_l, int Ho, int Wo,
                                   int act, float* y, float* channel_sums, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n >= 1 && channels >= 1 && (long long)n * channels <= 65535, "depthwise_conv: n*channels too large");
    dim3 grid(ceil_div(Wo, 32), ceil_div(Ho, 8), n * channels);
#define HEAL_DW(KK, ST)                                                                                       \
    if (ksize == KK && stride == ST) {                                                                        \
        k_depthwise<KK, ST><<<grid, 256, 0, s>>>(x, weight, bias, channels, H, W, Ho, Wo, pad_t, pad_l, act, y, \
                                                 channel_sums);                                          \
        HEAL_LAUNCH_CHECK();                                                                                  \
        return 0;                                                                                             \
    }
    HEAL_DW(3, 1) HEAL_DW(3, 2) HEAL_DW(5, 1) HEAL_DW(5, 2) HEAL_DW(7, 1)
#undef HEAL_DW
    return set_error("depthwise_conv: kernel %d stride %d is not instantiated", ksize, stride);
}

namespace heal {
// Squeeze-excite gate of an MBConv block in one launch: gate[n][c] = sigmoid(W2 silu(W1 m[n] + b1) + b2)[c] with m the
// spatial mean [n,C], W1 [S,C], W2 given TRANSPOSED [S,C] (S <= 64).  One 16-wave block per image; replaces the conv /
// SiLU / conv / sigmoid launches on 1x1 maps.  Kept deliberately small in code size: a fully unrolled variant was 5x
// faster in isolation and 5x SLOWER inside the pipeline (cold instruction fetch between hundreds of other kernels).
__global__ __launch_bounds__(1024) void k_se_gate(const float* __restrict__ mean, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2t,
                                                 const float* __restrict__ b2, int C, int S, float scale,
                                                 int tiles, float* __restrict__ gate) {
    // squeezed input m[c] = scale * sum_t mean[n][c][t]: tiles = 1, scale = 1 for a spatial mean; tiles = T, scale = 1/(Ho*Wo)
    // for the per-tile sums heal_depthwise_conv leaves (added here in tile order: deterministic).
    // grid = (n, channel blocks of 256): every block recomputes the S hidden units (S*C MACs, cheap) and finishes 256 output
    // channels with FOUR threads per channel (each a quarter of the hidden units, combined by two shuffles) -- one block per
    // image serialised S dependent loads per thread on 4 of the 256 CUs.
    __shared__ float hid[64];
    extern __shared__ float sm[];            // [C] squeezed input
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 1024) {
        const float* src = mean + ((size_t)n * C + c) * tiles;
        float acc = 0.f;
        for (int t = 0; t < tiles; ++t) acc += src[t];
        sm[c] = acc * scale;
    }
    __syncthreads();
    const float* m = sm;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int j = wave; j < S; j += 16) {  // one wave per hidden unit: coalesced row of W1, tree reduction
        const float* wr = w1 + (size_t)j * C;
        float acc = 0.f;
#pragma unroll 4
        for (int c = l; c < C; c += 64) acc = fmaf(wr[c], m[c], acc);
        acc = wave_sum(acc);
        if (l == 0) {
            const float v = acc + b1[j];
            hid[j] = v / (1.f + expf(-v));
        }
    }
    __syncthreads();
    // channel c = 256 * blockIdx.y + tid / 4, quarter q = tid % 4 of the hidden units (j = q, q + 4, ...): fixed order
    const int c = blockIdx.y * 256 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    float g = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int j = q; j < S; j += 4) g = fmaf(w2t[(size_t)j * C + c], hid[j], g);
    }
    g += __shfl_xor(g, 1, 64);
    g += __shfl_xor(g, 2, 64);
    if (c < C && q == 0) gate[(size_t)n * C + c] = 1.f / (1.f + expf(-(g + b2[c])));
}
}  // namespace heal

extern "C" int heal_se_gate(const float* mean, const float* w_reduce, const float* b_reduce, const float* w_expand_t,
                            const float* b_expand, int n, int channels, int squeezed, float scale, int tiles,
                            float* gate, void* stream) {
    HEAL_REQUIRE(tiles >= 1 && channels <= 12288, "se_gate: tiles must be >= 1 and channels <= 12288");
    HEAL_REQUIRE(n >= 1 && channels >= 1 && squeezed >= 1 && squeezed <= 64, "se_gate: squeezed channels must be in [1,64]");
    HEAL_REQUIRE(mean && w_reduce && b_reduce && w_expand_t && b_expand && gate, "se_gate: null pointer");
    heal::k_se_gate<<<dim3(n, ceil_div(channels, 256)), 1024, (size_t)channels * sizeof(float), (hipStream_t)stream>>>(
        mean, w_reduce, b_reduce, w_expand_t, b_expand, channels, squeezed, scale, tiles, gate);
    HEAL_LAUNCH_CHECK();
    return 0;
}

namespace heal {
// LayerNorm over the CHANNEL axis of an NCHW map (ConvNeXt block, feature_alignnet_modules.py:12-31,318-321: the
// reference permutes to NHWC and calls F.layer_norm).  One thread per pixel: consecutive threads read consecutive pixels
// of a channel plane (coalesced), two passes over the C planes (mean, then biased variance), eps inside the sqrt.
__global__ __launch_bounds__(256) void k_layernorm_nchw(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int C, int HW, float eps,
                                                       float* __restrict__ y) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* xi = x + (size_t)blockIdx.y * C * HW + p;
    float* yo = y + (size_t)blockIdx.y * C * HW + p;
    float mean = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) mean += xi[(size_t)c * HW];
    mean /= (float)C;
    float var = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float d = xi[(size_t)c * HW] - mean;
        var = fmaf(d, d, var);
    }
    const float inv = 1.f / sqrtf(var / (float)C + eps);
#pragma unroll 8
    for (int c = 0; c < C; ++c) yo[(size_t)c * HW] = (xi[(size_t)c * HW] - mean) * inv * gamma[c] + beta[c];
}

// Single-output pointwise convolution y[n][p] = b + sum_c w[c] x[n][c][p]: the occupancy heads of PyramidFusion
// (pyramid_fuse.py:89-91: nn.Conv2d(C, 1, kernel_size=1)).  A GEMM with one output row wastes a whole m-tile; this is a
// streaming reduction over the channel axis: 16 B of four pixels per thread and channel, weights broadcast through scalar
// loads, k ascending (the summation order of a plain dot product).  HBM-bound: 4 C bytes read per output pixel.
__global__ __launch_bounds__(256) void k_channel_dot(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int C, int HW4,
                                                    float* __restrict__ y) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const float4* __restrict__ xi = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.y * C * HW4 + q;
    const float b = bias ? bias[0] : 0.f;
    float4 acc = make_float4(b, b, b, b);
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float4 v = xi[(size_t)c * HW4];
        const float wc = w[c];
        acc.x = fmaf(wc, v.x, acc.x); acc.y = fmaf(wc, v.y, acc.y); acc.z = fmaf(wc, v.z, acc.z); acc.w = fmaf(wc, v.w, acc.w);
    }
    reinterpret_cast<float4*>(y)[(size_t)blockIdx.y * HW4 + q] = acc;
}
}  // namespace heal

extern "C" int heal_channel_dot(const float* x, const float* weight, const float* bias, int n, int channels, int HW,
                                float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && channels >= 1 && HW >= 4 && HW % 4 == 0 && n <= 65535, "channel_dot: bad shape (H*W %% 4 == 0)");
    HEAL_REQUIRE(x && weight && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "channel_dot: bad pointer");
    heal::k_channel_dot<<<dim3(ceil_div(HW / 4, 256), n), 256, 0, (hipStream_t)stream>>>(x, weight, bias, channels, HW / 4, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_layernorm_nchw(const float* x, const float* gamma, const float* beta, int n, int channels, int HW,
                                   float eps, float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && channels >= 1 && HW >= 1 && n <= 65535, "layernorm_nchw: bad shape");
    HEAL_REQUIRE(x && gamma && beta && y, "layernorm_nchw: null pointer");
    heal::k_layernorm_nchw<<<dim3(ceil_div(HW, 256), n), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, channels, HW, eps, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_upsample2x_bilinear(const float* x, int n, int channels, int H, int W, float* y, void* stream) {
    const long long total = (long long)n * channels * (2 * H) * (2 * W);
    if (total <= 0) return 0;
    k_upsample2x<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, H, W, total, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_grouped_conv3x3(const float* x, const float* weight, const float* bias, int n, int channels,
                                    int groups, int H, int W, int stride, int relu, float* y, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n >= 1 && channels >= 1 && groups >= 1 && channels % groups == 0, "grouped_conv3x3: bad channels");
    HEAL_REQUIRE(stride == 1 || stride == 2, "grouped_conv3x3: stride must be 1 or 2");
    const int cg = channels / groups;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    HEAL_REQUIRE((long long)n * groups <= 65535, "grouped_conv3x3: n*groups exceeds the grid limit");
    dim3 grid(ceil_div(Wo, 32), ceil_div(Ho, 8), n * groups);
#define HEAL_GC(CG, ST)                                                                                    \
    if (cg == CG && stride == ST) {                                                                        \
        k_grouped_conv3x3<CG, ST><<<grid, 256, 0, s>>>(x, weight, bias, channels, H, W, Ho, Wo, relu, y);   \
        HEAL_LAUNCH_CHECK();                                                                               \
        return 0;                                                                                          \
    }
    static const bool one_px = getenv("HEAL_GCONV_1PX") != nullptr;  // A/B switch for the packed-math kernel
    if (stride == 1 && !one_px) {
        dim3 g2(ceil_div(W, 64), ceil_div(H, 8), n * groups);
#define HEAL_GCP(CG)                                                                                       \
    if (cg == CG) {                                                                                        \
        k_grouped_conv3x3_pk<CG><<<g2, 256, 0, s>>>(x, weight, bias, channels, H, W, relu, y);              \
        HEAL_LAUNCH_CHECK();                                                                               \
        return 0;                                                                                          \
    }
        HEAL_GCP(4) HEAL_GCP(8) HEAL_GCP(16)
#undef HEAL_GCP
    }
    HEAL_GC(4, 1) HEAL_GC(4, 2) HEAL_GC(8, 1) HEAL_GC(8, 2) HEAL_GC(16, 1) HEAL_GC(16, 2)
#undef HEAL_GC
    return set_error("grouped_conv3x3: %d channels per group is not instantiated (4, 8, 16)", cg);
}

extern "C" int heal_bias_act(float* x, const float* bias, const float* residual, int n, int channels, int HW,
                             int relu, void* stream) {
    HEAL_REQUIRE(HW % 4 == 0, "bias_act: H*W must be a multiple of 4");
    const long long total4 = (long long)n * channels * (HW / 4);
    if (total4 <= 0) return 0;
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    k_bias_act<<<blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<float4*>(x), bias,
                                                        reinterpret_cast<const float4*>(residual), channels, HW / 4,
                                                        total4, relu);
    HEAL_LAUNCH_CHECK();
    return 0;
}
