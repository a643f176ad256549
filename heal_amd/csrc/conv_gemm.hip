// Dense NCHW convolution (3x3 padding 1, or 1x1; stride 1 | 2; round 6: also 7x7 padding 3 stride 2) as an implicit GEMM on the 128 x 128 x 32 core of linear.hip:
// v_mfma_f32_32x32x2_f32, 4 waves x (64 x 64), double-buffered LDS chunks, epilogue through LDS.
//
// Reference call sites: the stride-2 3x3 convolutions of the dense BEV stacks -- BaseBEVBackbone's stage heads
// (opencood/models/sub_modules/base_bev_backbone.py:49-74), the `shrink_header` DownsampleConv of HeterModelBaseline
// (downsample_conv.py:7-49; 384 -> 256 at 256^2 x 8 agents in BASELINE config 5: ONE launch of 232 GFLOP that round 2's
// 64 x 128-tile kernel on 16x16x4 ran at 65 TFLOP/s = 11 % of that step), BasicBlock / ResNeXt downsampling
// (resblock.py:18-64,160-165).
//
//   Y[co, p] = sum_{tap, ci} Wr[co, tap, ci] * X[ci, pixel(p) * s + tap - pad]      M = Cout, N = output pixels, K = taps * Cin
// A (weights, re-laid [Cout][tap][Cin] by the caller: a row's 32 input channels of one tap are contiguous) is staged exactly as
// linear.hip stages its operands; B (the input) is gathered per chunk = (tap, 32 input channels) x 128 output pixels of ONE
// image (TH rows x TW columns), 16 scalar loads per thread (a stride-2 row is every other float: nothing to vectorise; zero
// padding = clamped address + select), stored [k][pixel] so that the MFMA B fragment of step s is 32 consecutive floats.
// D[co][pixel]: a lane holds one pixel column of 4 + 4 + 4 + 4 output channels -- after the LDS transpose of the epilogue
// every thread writes 16-B pieces of an output row with bias / residual / ReLU applied.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int CG_BM = 128, CG_BN = 128, CG_BK = 32, CG_RSA = 36, CG_RSB = CG_BN + 4;

struct ConvGemmArgs {
    const float* x; const float* w; const float* bias; const float* res; float* y;
    int n, cin, cout, H, W, Ho, Wo, relu;
    int tw_log2;            // tile = (128 >> tw_log2) rows x (1 << tw_log2) columns of output pixels
    int tiles_x, tiles_y;
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256, 2) void k_conv_gemm(const ConvGemmArgs a) {
    constexpr int PAD = KS / 2, TAPS = KS * KS;
    __shared__ __attribute__((aligned(16))) float s_all[2 * CG_BM * CG_RSA + 2 * CG_BK * CG_RSB];
    float (*sA)[CG_BM * CG_RSA] = reinterpret_cast<float (*)[CG_BM * CG_RSA]>(s_all);
    float (*sB)[CG_BK * CG_RSB] = reinterpret_cast<float (*)[CG_BK * CG_RSB]>(s_all + 2 * CG_BM * CG_RSA);
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = l & 31, h = l >> 5;
    // block -> (cout tile fastest: the blocks that share an input patch are neighbours, pixel tile, image)
    const int m_tiles = a.cout / CG_BM;
    int b = blockIdx.x;
    const int tile_m = b % m_tiles; b /= m_tiles;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int img = b / a.tiles_y;
    const int TW = 1 << a.tw_log2, TH = CG_BN >> a.tw_log2;
    const int oy0 = ty * TH, ox0 = tx * TW, co0 = tile_m * CG_BM;

    // ---- staging roles ------------------------------------------------------------------------------------------------
    // A: 4 x 16 B per chunk, row (cout) = (tid + 256 i) / 8, 16-B column c4;  B: 16 floats per chunk, pixel p = tid % 128,
    // input channel (tid / 128) + 2 i
    const float* a_src[4];
    int sa_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i, row = c >> 3, c4 = c & 7;
        a_src[i] = a.w + (size_t)(co0 + row) * TAPS * a.cin + c4 * 4;
        sa_off[i] = row * CG_RSA + c4 * 4;
    }
    const int p = tid & 127, cb = tid >> 7;
    const int oy = oy0 + (p >> a.tw_log2), ox = ox0 + (p & (TW - 1));
    const int iy0 = oy * STRIDE - PAD, ix0 = ox * STRIDE - PAD;
    const bool p_ok = oy < a.Ho && ox < a.Wo;
    const size_t HW = (size_t)a.H * a.W;
    const float* x_img = a.x + (size_t)img * a.cin * HW;

    // staging registers: NAMED float4 values for the weights (an array of float4 written in one branch and read in another is
    // lowered through scratch memory: 4 scratch stores + loads per chunk in the round-3 kernel), and the padding mask of the
    // gathered activations applied at the LDS store (a select right behind the load waits for it in front of the MFMAs).
    float4 ra0, ra1, ra2, ra3;
    ra0 = ra1 = ra2 = ra3 = make_float4(0.f, 0.f, 0.f, 0.f);
    float rb[16];
    bool rb_ok = false;
#define HEAL_CG_LOAD(chunk_)                                                                                           \
    {                                                                                                                  \
        const int tap_ = (chunk_) / cpt, cc_ = (chunk_) - tap_ * cpt;                                                  \
        const int ky_ = tap_ / KS, kx_ = tap_ - ky_ * KS;                                                              \
        const size_t wo_ = (size_t)tap_ * a.cin + cc_ * CG_BK;                                                         \
        ra0 = *reinterpret_cast<const float4*>(a_src[0] + wo_);                                                        \
        ra1 = *reinterpret_cast<const float4*>(a_src[1] + wo_);                                                        \
        ra2 = *reinterpret_cast<const float4*>(a_src[2] + wo_);                                                        \
        ra3 = *reinterpret_cast<const float4*>(a_src[3] + wo_);                                                        \
        const int iy_ = iy0 + ky_, ix_ = ix0 + kx_;                                                                    \
        rb_ok = p_ok && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W;                                                \
        const float* src_ = x_img + (size_t)(cc_ * CG_BK + cb) * HW + (rb_ok ? (size_t)iy_ * a.W + ix_ : 0);           \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) rb[i] = src_[(size_t)(2 * i) * HW];                             \
    }
#define HEAL_CG_STORE(buf_)                                                                                            \
    {                                                                                                                  \
        *reinterpret_cast<float4*>(&sA[buf_][sa_off[0]]) = ra0;                                                        \
        *reinterpret_cast<float4*>(&sA[buf_][sa_off[1]]) = ra1;                                                        \
        *reinterpret_cast<float4*>(&sA[buf_][sa_off[2]]) = ra2;                                                        \
        *reinterpret_cast<float4*>(&sA[buf_][sa_off[3]]) = ra3;                                                        \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) sB[buf_][(cb + 2 * i) * CG_RSB + p] = rb_ok ? rb[i] : 0.f;      \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int cpt = a.cin / CG_BK;            // chunks per tap
    const int n_chunks = TAPS * cpt;
    HEAL_CG_LOAD(0)
    HEAL_CG_STORE(0)
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) HEAL_CG_LOAD(c + 1)   // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        float af[2][16], bf[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* q = &sA[buf][(wm * 64 + m * 32 + li) * CG_RSA + 16 * h];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(q + 4 * j);
                af[m][4 * j] = v.x; af[m][4 * j + 1] = v.y; af[m][4 * j + 2] = v.z; af[m][4 * j + 3] = v.w;
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float* q = &sB[buf][(16 * h) * CG_RSB + wn * 64 + n * 32 + li];
#pragma unroll
            for (int s = 0; s < 16; ++s) bf[n][s] = q[s * CG_RSB];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][s], bf[n][s], acc[m][n], 0, 0, 0);
        if (c + 1 < n_chunks) HEAL_CG_STORE(buf ^ 1)
        __syncthreads();
    }
#undef HEAL_CG_LOAD
#undef HEAL_CG_STORE

    // ---- epilogue through LDS: sC[cout row][pixel], 16-B pieces along the pixels ---------------------------------------------
    constexpr int CS = CG_BN + 4;
    static_assert(sizeof(s_all) >= (size_t)CG_BM * CS * 4, "epilogue tile must fit the operand rings");
    float* sC = s_all;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CS + wn * 64 + n * 32 + li] = acc[m][n][r];
    __syncthreads();
    const int c4 = tid & 31;                       // 16-B piece = pixels 4 c4 .. 4 c4 + 3 of the tile
    const int pp = c4 * 4, poy = oy0 + (pp >> a.tw_log2), pox = ox0 + (pp & (TW - 1));
    if (poy >= a.Ho || pox >= a.Wo) return;        // (Wo % 4 == 0 and TW % 4 == 0: a piece is inside or outside as a whole)
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const size_t pix = (size_t)poy * a.Wo + pox;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int row = (tid >> 5) + 8 * i, co = co0 + row;
        float4 v = *reinterpret_cast<const float4*>(&sC[row * CS + c4 * 4]);
        if (a.bias) { const float bv = a.bias[co]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
        const size_t o = ((size_t)img * a.cout + co) * HWo + pix;
        if (a.res) {
            const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(a.y + o) = v;
    }
}

}  // namespace heal

using namespace heal;

// weight: [Cout, ksize^2, Cin] (the caller re-lays [Cout, Cin, k, k] once: w.permute(0, 2, 3, 1)); Cout % 128 == 0, Cin % 32 == 0,
// Wo % 4 == 0.  y = act(conv(x) + bias (+ residual)).
extern "C" int heal_conv_gemm(const float* x, const float* weight_tap_major, const float* bias, const float* residual, int n,
                              int cin, int cout, int H, int W, int ksize, int stride, int relu, float* y, void* stream) {
    HEAL_REQUIRE(((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2)) || (ksize == 7 && stride == 2),
                 "conv_gemm: ksize 1 | 3 with stride 1 | 2, or the 7x7 / 2 stem (padding = ksize / 2)");
    HEAL_REQUIRE(cout % CG_BM == 0 && cin % CG_BK == 0 && n >= 1 && H >= 1 && W >= 1,
                 "conv_gemm: Cout must be a multiple of 128 and Cin of 32 (got %d -> %d)", cin, cout);
    ConvGemmArgs a;
    a.x = x; a.w = weight_tap_major; a.bias = bias; a.res = residual; a.y = y;
    a.n = n; a.cin = cin; a.cout = cout; a.H = H; a.W = W; a.relu = relu;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    HEAL_REQUIRE(a.Wo % 4 == 0, "conv_gemm: output width must be a multiple of 4 (got %d)", a.Wo);
    HEAL_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)weight_tap_major & 15) == 0 && (!residual || ((uintptr_t)residual & 15) == 0),
                 "conv_gemm: 16-B alignment");
    int twl = 6;                                  // 2 x 64 pixel tiles on wide maps, squarer tiles on narrow ones
    while (twl > 2 && (1 << twl) > a.Wo) --twl;
    a.tw_log2 = twl;
    const int TW = 1 << twl, TH = CG_BN >> twl;
    a.tiles_x = ceil_div(a.Wo, TW); a.tiles_y = ceil_div(a.Ho, TH);
    const long long blocks = (long long)(cout / CG_BM) * a.tiles_x * a.tiles_y * n;
    HEAL_REQUIRE(blocks < (1ll << 31), "conv_gemm: grid too large");
    hipStream_t s = (hipStream_t)stream;
#define HEAL_CG(KS_, ST_) HEAL_LAUNCH_EV((k_conv_gemm<KS_, ST_>), dim3((unsigned)blocks), dim3(256), 0, s, a)
    if (ksize == 7) HEAL_CG(7, 2);               // BevEncode's stem (lss_submodule.py:242: Conv2d(inC, 64, 7, stride 2, padding 3)), round 6
    else if (ksize == 3 && stride == 2) HEAL_CG(3, 2);
    else if (ksize == 3) HEAL_CG(3, 1);
    else if (stride == 2) HEAL_CG(1, 2);
    else HEAL_CG(1, 1);
#undef HEAL_CG
    HEAL_LAUNCH_CHECK();
    return 0;
}
