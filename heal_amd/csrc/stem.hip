// ResNet image stem: 7x7 stride-2 convolution (pad 3, <= 4 input channels -> 64) + folded BatchNorm + ReLU + 3x3 stride-2 max-pool
// (pad 1) in ONE kernel.
//
// Reference: the torchvision resnet101 stem of the Lift-Splat camera encoder, opencood/models/sub_modules/lss_submodule.py:153-161
// (construction) and :196-210 (conv1 -> bn1 -> relu -> maxpool); the same 7x7 / 2 stem without the pool opens BevEncode (:236-273).
// Through round 3 this was the last library convolution on the path (MIOpen `miopenSp3AsmConv...f3x2_stride2`, 86 us, preceded by
// a copy that made the x[:, :3] channel slice contiguous, followed by ATen `max_pool_forward_nchw`, 29 us).
//
// Formulation: implicit GEMM on v_mfma_f32_16x16x4_f32 -- M = 64 output channels, N = the 17 x 17 convolution pixels under an
// 8 x 8 tile of POOLED pixels (289, padded to 19 n-tiles), K = Cin * 49 (147 for RGB, padded to 37 k-steps).
//   * the block's input patch (39 x 39 per channel: stride 2 twice + both halos, zero padding resolved at staging time) goes
//     through LDS once (18 KB); the weight matrix in MFMA A-fragment order is read from L2, one k-step ahead of its use;
//   * wave w owns n-tiles w, w + 4, ... (5 | 5 | 5 | 4) x all four m-tiles: 20 accumulators; per k-step four A fragments (64
//     consecutive LDS words each) and five B fragments gathered from the patch at (channel, ky, kx) + (2 cy, 2 cx) -- the 16 pixel
//     lanes of a fragment step by 2 words and the k lanes of a 32-lane phase differ by an odd offset: conflict-free ds_read_b32;
//   * epilogue in two passes of 32 channels through the (then free) LDS: + bias, ReLU, pixels outside the convolution's output
//     set to -inf, then every thread takes the 3 x 3 maxima of 8 (channel, pooled pixel) pairs.  The convolution output
//     (38.5 MB for four 336 x 448 images) never reaches HBM.
// The image tensor is read IN PLACE: only the first Cin channels of rows `img_stride` floats apart (no contiguous copy of a
// channel slice).  Roofline: fp32 MFMA, 2 * 64 * 147 * 289 / 64 flops per pooled pixel.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int ST_PT = 8;                       // pooled tile side
constexpr int ST_CT = 2 * ST_PT + 1;           // convolution tile side (17)
constexpr int ST_IT = 2 * ST_CT + 5;           // input patch side (39)
constexpr int ST_NPIX = ST_CT * ST_CT;         // 289
constexpr int ST_NT = (ST_NPIX + 15) / 16;     // 19 n-tiles
constexpr int ST_NTW = (ST_NT + 3) / 4;        // n-tiles per wave (5)
constexpr int ST_CS = ST_NPIX + 2;             // epilogue row stride (291: odd)

template <int KS>   // k-steps: ceil(Cin * 49 / 4)
__global__ __launch_bounds__(256, 4) void k_stem7x7(const float* __restrict__ x, long long img_stride, int Cin, int H, int W,
                                                   const float* __restrict__ wfrag /*[4][KS][64]*/,
                                                   const float* __restrict__ bias, int Hc, int Wc, int Hp, int Wp,
                                                   int tiles_x, int pool, float* __restrict__ y) {
    // LDS holds the input patch only (and later the epilogue tile): the A fragments (weights) come straight from L2 -- 64
    // consecutive words per (m-tile, k-step), requested one k-step ahead.  With the 38 KB weight matrix staged in LDS as well only
    // two blocks fit a CU and the 616 blocks of four 336 x 448 images ran as two rounds (95 us); now four-plus blocks are resident.
    constexpr int PATCH = ((KS * 4 + 48) / 49) * ST_IT * ST_IT;              // ceil(K / 49) input channels
    constexpr int S_ALL = PATCH > 32 * ST_CS ? PATCH : 32 * ST_CS;
    __shared__ __attribute__((aligned(16))) float s_all[S_ALL];
    float* sP = s_all;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63, lk = l >> 4, ln = l & 15;
    const int n = blockIdx.y;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    // pool: the tile is 8 x 8 pooled pixels over convolution rows 2 py0 - 1 .. 2 py0 + 15; no pool: a 17 x 17 convolution tile
    const int py0 = tyi * ST_PT, px0 = txi * ST_PT;
    const int cy0 = pool ? 2 * py0 - 1 : tyi * ST_CT, cx0 = pool ? 2 * px0 - 1 : txi * ST_CT;
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;

    // the input patch -> LDS
    const float* __restrict__ xin = x + (size_t)n * img_stride;
    for (int e = tid; e < Cin * ST_IT * ST_IT; e += 256) {
        const int c = e / (ST_IT * ST_IT), r = e - c * (ST_IT * ST_IT), py = r / ST_IT, px = r - py * ST_IT;
        const int gy = iy0 + py, gx = ix0 + px;
        sP[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? xin[((size_t)c * H + gy) * W + gx] : 0.f;
    }
    __syncthreads();

    // B-fragment pixel offsets of this lane: n-tile nt = wave + 4 j, pixel nt * 16 + ln -> (2 cy, 2 cx) inside the patch
    int poff[ST_NTW];
#pragma unroll
    for (int j = 0; j < ST_NTW; ++j) {
        const int pixel = min((wave + 4 * j) * 16 + ln, ST_NPIX - 1);       // padding lanes read a valid word, results unused
        const int cy = pixel / ST_CT, cx = pixel - cy * ST_CT;
        poff[j] = 2 * cy * ST_IT + 2 * cx;
    }
    f32x4 acc[4][ST_NTW];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < ST_NTW; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int K = Cin * 49;
    float a_nxt[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = wfrag[(mt * KS) * 64 + l];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = min(4 * ks + lk, K - 1);                              // k >= K: the weight fragment holds zeros
        const int c = k / 49, r = k - c * 49, ky = r / 7, kx = r - ky * 7;
        const int koff = c * (ST_IT * ST_IT) + ky * ST_IT + kx;
        float a[4], b[ST_NTW];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = a_nxt[mt];
        if (ks + 1 < KS) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a_nxt[mt] = wfrag[(mt * KS + ks + 1) * 64 + l];
        }
#pragma unroll
        for (int j = 0; j < ST_NTW; ++j) b[j] = sP[koff + poff[j]];
#pragma unroll
        for (int j = 0; j < ST_NTW; ++j) {
            if (wave + 4 * j < ST_NT) {                                     // wave 3 owns four n-tiles
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[j], acc[mt][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                                        // operand tiles are free from here on

    // epilogue: two passes of 32 channels through LDS.  D[row = 4 lk + r][col = ln] of (mt, j): channel mt * 16 + 4 lk + r,
    // convolution pixel (wave + 4 j) * 16 + ln.
    float* sC = s_all;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int mt = 2 * pass + mh;
#pragma unroll
            for (int j = 0; j < ST_NTW; ++j) {
                const int pixel = (wave + 4 * j) * 16 + ln;
                if (wave + 4 * j < ST_NT && pixel < ST_NPIX) {
                    const int cy = pixel / ST_CT, cx = pixel - cy * ST_CT;
                    const bool in = cy0 + cy >= 0 && cy0 + cy < Hc && cx0 + cx >= 0 && cx0 + cx < Wc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mt * 16 + 4 * lk + r;
                        const float v = fmaxf(acc[mt][j][r] + (bias ? bias[co] : 0.f), 0.f);
                        sC[(mh * 16 + 4 * lk + r) * ST_CS + pixel] = in ? v : -INFINITY;
                    }
                }
            }
        }
        __syncthreads();
        if (pool) {
#pragma unroll
            for (int i = 0; i < 32 * ST_PT * ST_PT / 256; ++i) {
                const int o = tid + 256 * i, cl = o >> 6, pp = o & 63, py = pp >> 3, px = pp & 7;
                if (py0 + py < Hp && px0 + px < Wp) {
                    const float* s = sC + cl * ST_CS + (2 * py) * ST_CT + 2 * px;
                    float m = fmaxf(fmaxf(s[0], s[1]), s[2]);
                    m = fmaxf(m, fmaxf(fmaxf(s[ST_CT], s[ST_CT + 1]), s[ST_CT + 2]));
                    m = fmaxf(m, fmaxf(fmaxf(s[2 * ST_CT], s[2 * ST_CT + 1]), s[2 * ST_CT + 2]));
                    y[(((size_t)n * 64 + pass * 32 + cl) * Hp + py0 + py) * Wp + px0 + px] = m;
                }
            }
        } else {
            for (int o = tid; o < 32 * ST_NPIX; o += 256) {
                const int cl = o / ST_NPIX, pixel = o - cl * ST_NPIX, cy = pixel / ST_CT, cx = pixel - cy * ST_CT;
                if (cy0 + cy < Hc && cx0 + cx < Wc)
                    y[(((size_t)n * 64 + pass * 32 + cl) * Hc + cy0 + cy) * Wc + cx0 + cx] = sC[cl * ST_CS + pixel];
            }
        }
        __syncthreads();
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_stem7x7(const float* x, long long image_stride, int n, int cin, int H, int W, const float* weight_frag,
                            const float* bias, int pool, float* y, void* stream) {
    HEAL_REQUIRE(x && weight_frag && y, "stem7x7: null pointer");
    HEAL_REQUIRE(n >= 1 && n <= 65535 && cin >= 1 && cin <= 4 && H >= 1 && W >= 1, "stem7x7: 1..4 input channels (got %d)", cin);
    HEAL_REQUIRE(image_stride >= (long long)cin * H * W, "stem7x7: image stride smaller than Cin*H*W");
    HEAL_REQUIRE(((uintptr_t)weight_frag & 15) == 0, "stem7x7: weight fragments must be 16-B aligned");
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;          // (H + 6 - 7) / 2 + 1
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;          // (Hc + 2 - 3) / 2 + 1
    const int tiles_x = pool ? ceil_div(Wp, ST_PT) : ceil_div(Wc, ST_CT);
    const int tiles_y = pool ? ceil_div(Hp, ST_PT) : ceil_div(Hc, ST_CT);
    const dim3 grid(tiles_x * tiles_y, n);
    hipStream_t s = (hipStream_t)stream;
    const int ks = (cin * 49 + 3) / 4;
#define HEAL_STEM(KS_)                                                                                                    \
    if (ks == KS_) {                                                                                                      \
        HEAL_LAUNCH_EV(k_stem7x7<KS_>, grid, dim3(256), 0, s, x, image_stride, cin, H, W, weight_frag, bias, Hc, Wc, Hp, Wp, \
                       tiles_x, pool ? 1 : 0, y);                                                                         \
    }
    HEAL_STEM(13) HEAL_STEM(25) HEAL_STEM(37) HEAL_STEM(49)
#undef HEAL_STEM
    HEAL_LAUNCH_CHECK();
    return 0;
}
