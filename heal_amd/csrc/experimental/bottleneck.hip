// K7b -- one fused kernel for a whole ResNeXt bottleneck block of the pyramid backbone:
//     y = relu( conv3_1x1( relu( gconv2_3x3( relu( conv1_1x1(x) + b1 ) ) + b2 ) ) + b3 + x )
// (opencood/models/sub_modules/resblock.py:100-122 with the BatchNorms folded into weights/biases,
//  stride 1, no downsample, expansion 1, 32 groups: the 13 non-first blocks of PyramidFusion's
//  ResNeXt stages, pyramid_fuse.py:71-79.)
//
// Un-fused, a block is 5 passes over [n, C..2C, H, W] maps (GEMM, bias/ReLU, grouped conv, GEMM,
// bias/residual/ReLU): ~5.5x the compulsory HBM traffic of "read x once, write y once".  Here a workgroup
// owns a spatial tile; the 2C-wide intermediate never leaves the CU:
//   phase 1  conv1 on tile + halo as a [64-channel chunk x Cin] x [Cin x pixels] GEMM on the fp32 matrix
//            cores (v_mfma_f32_16x16x4_f32), bias + ReLU, zero outside the image (= conv2's zero padding)
//   phase 2  grouped 3x3 on the chunk (VALU stencil out of LDS), bias + ReLU
//   phase 3  conv3 partial product  acc[Cout x tile] += W3[:, chunk] x t2   (MFMA, accumulators stay in
//            registers across the width chunks)
//   epilogue bias + residual (x is still in LDS) + ReLU, store.
// Weight matrices arrive pre-arranged in MFMA A-fragment order (one coalesced 256 B load per fragment).
#include "../common.h"
#include "../../../include/heal_amd.h"
#include "../../../include/heal_amd_experimental.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CIN, int TW, int TH>
struct BnCfg {
    static constexpr int WIDTH = 2 * CIN;
    static constexpr int CG = WIDTH / 32;          // channels per group
    static constexpr int PW = TW + 2, PH = TH + 2;
    static constexpr int P1 = PW * PH;             // tile + halo pixels
    static constexpr int P1P = (P1 + 15) / 16 * 16;
    static constexpr int XROW = (P1P % 32 == 16) ? P1P : P1P + 16;   // row stride == 16 mod 32 floats
    static constexpr int P0 = TW * TH;             // output pixels (multiple of 16)
    static constexpr int T2ROW = (P0 % 32 == 16) ? P0 : P0 + 16;
    static constexpr int WC = 64;                  // width channels per chunk
    static constexpr int NCHUNK = WIDTH / WC;
    static constexpr int KS1 = CIN / 4;            // k-steps of phase 1
    static constexpr int NT1 = P1P / 16;           // n-tiles of phase 1
    static constexpr int MT3 = CIN / 16;           // m-tiles of phase 3 (Cout = CIN)
    static constexpr int MT3W = MT3 / 4;           // per wave
    static constexpr int NT3 = P0 / 16;
    static constexpr int COB = CG < 8 ? CG : 8;    // output channels per phase-2 work item
    static constexpr size_t LDS_FLOATS = (size_t)CIN * XROW + (size_t)WC * XROW + (size_t)WC * T2ROW;
};

template <int CIN, int TW, int TH>
__global__ __launch_bounds__(256) void k_bottleneck(const float* __restrict__ x,
                                                   const float* __restrict__ w1f, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ w3f, const float* __restrict__ b3,
                                                   int H, int W, float* __restrict__ y) {
    using C = BnCfg<CIN, TW, TH>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sx = lds;                                   // [CIN][XROW]   x tile + halo
    float* st1 = sx + (size_t)CIN * C::XROW;           // [WC][XROW]    conv1 chunk
    float* st2 = st1 + (size_t)C::WC * C::XROW;        // [WC][T2ROW]   grouped-conv chunk

    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t HWs = (size_t)H * W;
    const float* xin = x + (size_t)n * CIN * HWs;

    // ---- stage x (tile + halo, zero outside the image and in the padding columns) ----------------------
    for (int e = threadIdx.x; e < CIN * C::P1P; e += 256) {
        const int c = e / C::P1P, p = e - c * C::P1P;
        float v = 0.f;
        if (p < C::P1) {
            const int py = p / C::PW, px = p - py * C::PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = xin[(size_t)c * HWs + (size_t)gy * W + gx];
        }
        sx[c * C::XROW + p] = v;
    }
    __syncthreads();

    f32x4 acc3[C::MT3W][C::NT3];
#pragma unroll
    for (int a = 0; a < C::MT3W; ++a)
#pragma unroll
        for (int b = 0; b < C::NT3; ++b) acc3[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
        // ---- phase 1: t1[chunk rows][halo pixels] = relu(W1 x + b1), wave w owns 16 of the 64 rows -------
        {
            const int mt = chunk * 4 + wave;  // m-tile in units of 16 width channels
            float a[C::KS1];
#pragma unroll
            for (int ks = 0; ks < C::KS1; ++ks) a[ks] = w1f[((size_t)mt * C::KS1 + ks) * 64 + l];
            float bias[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = b1[mt * 16 + (l >> 4) * 4 + r];
            for (int nt = 0; nt < C::NT1; ++nt) {
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < C::KS1; ++ks) {
                    const float b = sx[(ks * 4 + (l >> 4)) * C::XROW + nt * 16 + (l & 15)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b, acc, 0, 0, 0);
                }
                // C/D layout: col (pixel) = l & 15, row (channel) = (l >> 4) * 4 + r
                const int p = nt * 16 + (l & 15);
                bool inside = false;
                if (p < C::P1) {
                    const int py = p / C::PW, px = p - py * C::PW;
                    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                    inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = wave * 16 + (l >> 4) * 4 + r;
                    st1[ch * C::XROW + p] = inside ? fmaxf(acc[r] + bias[r], 0.f) : 0.f;
                }
            }
        }
        __syncthreads();
        // ---- phase 2: grouped 3x3 over the chunk's 64 channels -> t2[64][P0] ------------------------------
        // output-channel blocks are dealt to waves (wave-uniform -> the 3x3 weights come through scalar
        // loads), lanes run over the tile's pixels
        for (int obi = wave; obi < C::WC / C::COB; obi += 4) {
            const int ob = __builtin_amdgcn_readfirstlane(obi);
            const int co0 = ob * C::COB;                 // first output channel (within the chunk)
            const int g0 = (co0 / C::CG) * C::CG;        // first channel of its group (within the chunk)
            const int wch = chunk * C::WC + co0;         // global width channel of co0
            const float* __restrict__ wblk = w2 + (size_t)wch * C::CG * 9;
            for (int p = l; p < C::P0; p += 64) {
                const int py = p / TW, px = p - py * TW;
                float acc[C::COB];
#pragma unroll
                for (int co = 0; co < C::COB; ++co) acc[co] = b2[wch + co];
#pragma unroll
                for (int ci = 0; ci < C::CG; ++ci) {
                    float v[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) v[k] = st1[(g0 + ci) * C::XROW + (py + k / 3) * C::PW + px + k % 3];
#pragma unroll
                    for (int co = 0; co < C::COB; ++co) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) acc[co] = fmaf(v[k], wblk[(co * C::CG + ci) * 9 + k], acc[co]);
                    }
                }
#pragma unroll
                for (int co = 0; co < C::COB; ++co) st2[(co0 + co) * C::T2ROW + p] = fmaxf(acc[co], 0.f);
            }
        }
        __syncthreads();
        // ---- phase 3: acc3 += W3[:, chunk] x t2 --------------------------------------------------------------
#pragma unroll
        for (int mi = 0; mi < C::MT3W; ++mi) {
            const int mt = wave * C::MT3W + mi;
            float a[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                a[ks] = w3f[((size_t)mt * (C::WIDTH / 4) + chunk * 16 + ks) * 64 + l];
#pragma unroll
            for (int nt = 0; nt < C::NT3; ++nt) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const float b = st2[(ks * 4 + (l >> 4)) * C::T2ROW + nt * 16 + (l & 15)];
                    acc3[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b, acc3[mi][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // st1 / st2 are rewritten by the next chunk
    }
    // ---- epilogue: + b3 + residual, ReLU, store ------------------------------------------------------------
    float* yout = y + (size_t)n * CIN * HWs;
#pragma unroll
    for (int mi = 0; mi < C::MT3W; ++mi) {
        const int mt = wave * C::MT3W + mi;
#pragma unroll
        for (int nt = 0; nt < C::NT3; ++nt) {
            const int p = nt * 16 + (l & 15);
            const int py = p / TW, px = p - py * TW;
            const int gy = y0 + py, gx = x0 + px;
            if (gy < H && gx < W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = mt * 16 + (l >> 4) * 4 + r;
                    const float res = sx[co * C::XROW + (py + 1) * C::PW + px + 1];
                    yout[(size_t)co * HWs + (size_t)gy * W + gx] = fmaxf(acc3[mi][nt][r] + b3[co] + res, 0.f);
                }
            }
        }
    }
}

template <int CIN, int TW, int TH>
static int launch_bottleneck(const float* x, const float* w1f, const float* b1, const float* w2, const float* b2,
                             const float* w3f, const float* b3, int n, int H, int W, float* y, hipStream_t s) {
    using C = BnCfg<CIN, TW, TH>;
    const size_t lds_bytes = C::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        HEAL_HIP(hipFuncSetAttribute((const void*)k_bottleneck<CIN, TW, TH>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    dim3 grid(ceil_div(W, TW), ceil_div(H, TH), n);
    k_bottleneck<CIN, TW, TH><<<grid, 256, lds_bytes, s>>>(x, w1f, b1, w2, b2, w3f, b3, H, W, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}

}  // namespace heal

using namespace heal;

extern "C" int heal_resnext_bottleneck(const float* x, const float* w1_frag, const float* b1, const float* w2,
                                       const float* b2, const float* w3_frag, const float* b3, int n, int channels,
                                       int H, int W, float* y, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n >= 1 && n <= 65535 && H >= 1 && W >= 1, "resnext_bottleneck: bad shape");
    switch (channels) {
        case 64: return launch_bottleneck<64, 16, 8>(x, w1_frag, b1, w2, b2, w3_frag, b3, n, H, W, y, s);
        case 128: return launch_bottleneck<128, 8, 8>(x, w1_frag, b1, w2, b2, w3_frag, b3, n, H, W, y, s);
        case 256: return launch_bottleneck<256, 8, 4>(x, w1_frag, b1, w2, b2, w3_frag, b3, n, H, W, y, s);
        default: return set_error("resnext_bottleneck: channels must be 64, 128 or 256 (got %d)", channels);
    }
}
