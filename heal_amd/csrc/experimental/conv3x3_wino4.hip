// Winograd F(4x4, 3x3) on the fp32 matrix cores (dense 3x3 convolution, stride 1, padding 1, NCHW).
//
// Reference call sites: the stride-1 3x3 convolutions of the dense BEV stacks -- BasicBlock / DoubleConv / BaseBEVBackbone
// (opencood/models/sub_modules/resblock.py:18-64, downsample_conv.py:7-49, base_bev_backbone.py:49-74).
//
// heal_conv3x3_winograd (F(2x2,3x3), conv3x3.hip) executes 16/36 of the direct convolution's multiplications; its MFMA phase is
// still three quarters of its time (ablation in DESIGN 8).  F(4x4,3x3) executes 36/144 = 1/4 of them:
//     Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A,     d: 6x6 input window (stride 4), g: 3x3 filter, Y: 4x4 outputs
// i.e. 36 independent GEMMs  M[xi][co, tile] = sum_ci U[xi][co, ci] V[xi][ci, tile]  over the transform positions xi.
//   * block = 8 waves = 32 output channels x (16 x 32 output pixels = 4 x 8 tiles of 4x4); wave w owns the nine positions
//     xi = 9 (w & 3) .. + 8 for the 16 channels (w >> 2): 9 x 2 accumulator tiles = 72 registers;
//   * per chunk of 16 input channels: the 18 x 34 input patch goes global -> registers (one chunk ahead) -> LDS; thread
//     (ci, tile) forms V = B^T d B (6x6, two passes of 6-point transforms) and writes 36 values to sV[xi][ci][tile]; then every
//     wave issues 9 xi x 4 k-steps x 2 n-tiles MFMAs (v_mfma_f32_16x16x4_f32) with B fragments from sV (row stride 48 words:
//     conflict-free) and A fragments (U) straight from L2 in a lane-major pre-laid order (nine 16-B loads per lane per chunk,
//     requested right after the MFMAs of the previous chunk);
//   * LDS layouts: patch channel stride 673 words (= 1 mod 32) and lanes = (4 channels x 8 tile columns): the 36 scalar window
//     reads of the transform are conflict-free although neighbouring tiles are 4 words apart;
//   * epilogue: four passes of 8 output channels through LDS (sM[xi][co][tile], aliasing sV); thread (co, tile, half) gathers its
//     36 transform-domain sums, applies two rows of A^T . A, bias (+ residual) (+ ReLU) and stores 2 x 4 pixels as 16-B rows.
// STATUS (round 3): correct on every test shape, NOT faster than F(2x2,3x3): 0.80 - 1.03x at the shapes of the two BASELINE scenes
// (profiles/r03_wino_f44_vs_f22.json).  With 8 waves the accumulators only fit for 32 output channels per block, so every input
// tile is transformed twice as often as in the F(2x2,3x3) kernel (64 channels per block), the kernel sits at the 256-register
// limit (the compiler then issues every LDS fragment read right in front of its MFMA pair and sinks the patch loads to their use),
// and what the smaller MFMA count saves goes into the transform.  Variants measured and dropped: U through a three-register ring
// + patch loads pinned above the MFMAs (0.71 - 0.91x, spills).  Opt-in: HEAL_C3_ALGO=winograd4.
// Arithmetic: fp32 throughout.  The transform constants (up to 8) cost accuracy against F(2x2,3x3): ~2e-6 median, 2e-5 worst
// relative to the output scale at Cin = 384 (measured against float64), inside the 1e-3 feature tolerance and the 1e-4 the
// kernel tests use.
#include <stdlib.h>
#include "../common.h"
#include "../../../include/heal_amd.h"
#include "../../../include/heal_amd_experimental.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int W4_KC = 16;                 // input channels per chunk
constexpr int W4_TR = 16, W4_TC = 32;     // output pixels per block
constexpr int W4_PR = W4_TR + 2, W4_PC = W4_TC + 2;   // patch rows / columns
constexpr int W4_PROW = 36;               // patch row stride (words)
constexpr int W4_PCI = 673;               // patch channel stride: >= 18 * 36 and = 1 (mod 32)
constexpr int W4_PE = W4_PR * W4_PC;      // 612 patch elements per channel
constexpr int W4_NP = (W4_KC * W4_PE + 511) / 512;    // 20 patch elements staged per thread
constexpr int W4_PDUMMY = 512;            // landing zone of the staging slots beyond the patch
constexpr int W4_VROW = 48;               // sV row stride: 32 tiles + 16 (k-rows lk, lk + 1 of a B fragment in disjoint banks)
constexpr int W4_MROW = 36;               // sM row stride (32 tiles + 4)
constexpr int W4_SP = W4_KC * W4_PCI + W4_PDUMMY;
constexpr int W4_SV = 36 * W4_KC * W4_VROW;
static_assert(W4_SV >= 36 * 8 * W4_MROW, "the epilogue buffer aliases sV");
static_assert((W4_SP + W4_SV) * 4 <= 160 * 1024, "LDS budget");

// one 6-point input transform: t = B^T d for  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w4_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                      float& t0, float& t1, float& t2, float& t3, float& t4, float& t5) {
    t0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);     // rows 1, 2: (d4 - 4 d2) +- (d3 - 4 d1)
    t1 = a + b;
    t2 = a - b;
    const float c = d4 - d2, e = 2.f * (d3 - d1);                   // rows 3, 4: (d4 - d2) +- 2 (d3 - d1)
    t3 = c + e;
    t4 = c - e;
    t5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}

__global__ __launch_bounds__(512) void k_conv3x3_wino4(const float* __restrict__ x, const float4* __restrict__ ufrag,
                                                      const float* __restrict__ bias, const float* __restrict__ res, int Cin,
                                                      int nchunks, int Cout, int H, int W, int tiles_x, int relu,
                                                      float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float smem[W4_SP + W4_SV];
    float* sP = smem;
    float* sV = smem + W4_SP;
    float* sM = sV;

    // block order as k_conv3x3_wino: tile fastest, image next, Cout block slowest (the blocks in flight share one U slab)
    const Block3 bk = xcd_block();
    const int mb = bk.z, n = bk.y;
    const int tyb = bk.x / tiles_x, txb = bk.x - tyb * tiles_x;
    const int oy0 = tyb * W4_TR, ox0 = txb * W4_TC;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const size_t HW = (size_t)H * W;
    const float* __restrict__ xin = x + (size_t)n * Cin * HW;

    // patch staging: element e = tid + 512 j of the [16][18][34] patch; loads unconditional (clamped address), zero padding and
    // the channel tail applied when the value goes to LDS (see k_conv3x3_wino for why).  The element's (channel, row, column)
    // and its in-image flag are packed into 16 bits, two elements per register: recomputing them from e with constant
    // divisions cost as many vector-ALU cycles per chunk as the transform itself, and 2 x 20 offset registers do not fit.
    //   bits 0-5 px (63: slot beyond the patch), 6-10 py, 11-14 ci, 15 inside the image
    unsigned pk[W4_NP / 2];
#pragma unroll
    for (int j = 0; j < W4_NP; ++j) {
        const int e = threadIdx.x + 512 * j;
        const int ci = e / W4_PE, rem = e - ci * W4_PE, py = rem / W4_PC, px = rem - py * W4_PC;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        const bool in_patch = e < W4_KC * W4_PE;
        const bool ok = in_patch && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const unsigned f = in_patch ? ((unsigned)px | ((unsigned)py << 6) | ((unsigned)ci << 11) | (ok ? 0x8000u : 0u)) : 63u;
        if (j & 1) pk[j >> 1] |= f << 16;
        else pk[j >> 1] = f;
    }
    const int base_off = (oy0 - 1) * W + ox0 - 1;
    float pst[W4_NP];
#define HEAL_W4_LOAD_PATCH(c_)                                                                                         \
    {                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < W4_NP; ++j) {                                                            \
            const unsigned f_ = (pk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;                                              \
            const int px_ = f_ & 63, py_ = (f_ >> 6) & 31, ci_ = (f_ >> 11) & 15;                                      \
            const int ch_ = min((c_) * W4_KC + ci_, Cin - 1);                                                          \
            const int off_ = (f_ & 0x8000u) ? base_off + py_ * W + px_ : 0;                                            \
            pst[j] = xin[(size_t)ch_ * HW + off_];                                                                     \
        }                                                                                                              \
    }
#define HEAL_W4_STORE_PATCH(c_)                                                                                        \
    {                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < W4_NP; ++j) {                                                            \
            const unsigned f_ = (pk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;                                              \
            const int px_ = f_ & 63, py_ = (f_ >> 6) & 31, ci_ = (f_ >> 11) & 15;                                      \
            const bool ok_ = (f_ & 0x8000u) && (c_) * W4_KC + ci_ < Cin;                                               \
            const int lds_ = px_ != 63 ? ci_ * W4_PCI + py_ * W4_PROW + px_ : W4_KC * W4_PCI + (int)(threadIdx.x & (W4_PDUMMY - 1)); \
            sP[lds_] = ok_ ? pst[j] : 0.f;                                                                             \
        }                                                                                                              \
    }
    // U fragments of this wave: 36 floats per lane per chunk, lane-major: [mb][chunk][wave][lane][xi_i][ks]
    const float4* __restrict__ ubase = ufrag + (((size_t)mb * nchunks * 8 + wave) * 64 + l) * 9;
    float4 ua[9];
#define HEAL_W4_LOAD_U(c_)                                                                                             \
    {                                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 9; ++q) ua[q] = ubase[(size_t)(c_) * 8 * 64 * 9 + q];                    \
    }

    f32x4 acc[9][2];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transform role: wave -> tile row (wave & 3) and channel half (wave >> 2); lane -> channel 4 (l >> 5) + (l & 3), tile
    // column (l >> 2) & 7: a 32-lane LDS phase covers 4 channels x 8 tile columns = 32 different banks
    const int tty = wave & 3, ttx = (l >> 2) & 7, tci = (wave >> 2) * 8 + (l >> 5) * 4 + (l & 3);
    const float* __restrict__ dsrc = sP + tci * W4_PCI + (4 * tty) * W4_PROW + 4 * ttx;
    float* __restrict__ vdst = sV + tci * W4_VROW + tty * 8 + ttx;
    // MFMA role: positions 9 (wave & 3) .. + 8, output channels 16 (wave >> 2) .. + 15 of the block's 32
    const int xg = wave & 3, mt = wave >> 2;

    HEAL_W4_LOAD_PATCH(0)
    HEAL_W4_LOAD_U(0)
    HEAL_W4_STORE_PATCH(0)
    for (int c = 0; c < nchunks; ++c) {
        const int cn = min(c + 1, nchunks - 1);
        lds_barrier();                            // patch(c) is in LDS; every wave is done with sV of chunk c-1
        {   // V = B^T d B: rows first (t = B^T d, column by column), then columns (v = t B, row by row)
            float t[6][6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                w4_bt(dsrc[0 * W4_PROW + j], dsrc[1 * W4_PROW + j], dsrc[2 * W4_PROW + j], dsrc[3 * W4_PROW + j],
                      dsrc[4 * W4_PROW + j], dsrc[5 * W4_PROW + j], t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], t[5][j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float v0, v1, v2, v3, v4, v5;
                w4_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], v0, v1, v2, v3, v4, v5);
                vdst[((6 * i + 0) * W4_KC) * W4_VROW] = v0;
                vdst[((6 * i + 1) * W4_KC) * W4_VROW] = v1;
                vdst[((6 * i + 2) * W4_KC) * W4_VROW] = v2;
                vdst[((6 * i + 3) * W4_KC) * W4_VROW] = v3;
                vdst[((6 * i + 4) * W4_KC) * W4_VROW] = v4;
                vdst[((6 * i + 5) * W4_KC) * W4_VROW] = v5;
            }
        }
        lds_barrier();                            // sV(c) complete; sP free
        HEAL_W4_LOAD_PATCH(cn)                    // in flight under the MFMAs (requested here, not before the transform: its 36
                                                  // temporaries and the 20 staging registers would not fit the 256-register budget)
#pragma unroll
        for (int xi_i = 0; xi_i < 9; ++xi_i) {
            const float* __restrict__ vb = sV + ((9 * xg + xi_i) * W4_KC + lk) * W4_VROW + ln;
            const float a_[4] = {ua[xi_i].x, ua[xi_i].y, ua[xi_i].z, ua[xi_i].w};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float b0 = vb[ks * 4 * W4_VROW], b1 = vb[ks * 4 * W4_VROW + 16];
                acc[xi_i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ks], b0, acc[xi_i][0], 0, 0, 0);
                acc[xi_i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[ks], b1, acc[xi_i][1], 0, 0, 0);
            }
        }
        HEAL_W4_LOAD_U(cn)                        // next chunk's U: lands during the barrier + transform that follow
        HEAL_W4_STORE_PATCH(cn)                   // sP is free since the second barrier
    }
#undef HEAL_W4_LOAD_PATCH
#undef HEAL_W4_STORE_PATCH
#undef HEAL_W4_LOAD_U

    // Epilogue: four passes of 8 output channels (rows lk >> 1 == p & 1 of the m-tile p >> 1) through sM[xi][8][tile]
    const size_t HWo = HW;
    float* __restrict__ yout = y + (size_t)n * Cout * HWo;
    const float* __restrict__ rin = res ? res + (size_t)n * Cout * HWo : nullptr;
    const int eh = threadIdx.x >> 8, ec = (threadIdx.x >> 5) & 7, etile = threadIdx.x & 31;   // row half, channel, tile
    const int ety = etile >> 3, etx = etile & 7;
    const int oy = oy0 + 4 * ety + 2 * eh, ox = ox0 + 4 * etx;
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        lds_barrier();                            // sV (first pass) / the previous pass no longer read
        if (mt == (p >> 1) && (lk >> 1) == (p & 1)) {
#pragma unroll
            for (int xi_i = 0; xi_i < 9; ++xi_i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sM[((9 * xg + xi_i) * 8 + (lk & 1) * 4 + r) * W4_MROW + nt * 16 + ln] = acc[xi_i][nt][r];
        }
        lds_barrier();
        const int co = mb * 32 + p * 8 + ec;
        // rows 2 eh, 2 eh + 1 of  A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        float u[2][6];   // u = (A^T M) rows 2 eh, 2 eh + 1
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = sM[((6 * i + j) * 8 + ec) * W4_MROW + etile];
            const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            if (eh == 0) {
                u[0][j] = (m[0] + s12) + s34;
                u[1][j] = fmaf(2.f, d34, d12);
            } else {
                u[0][j] = fmaf(4.f, s34, s12);
                u[1][j] = fmaf(8.f, d34, d12) + m[5];
            }
        }
        if (co >= Cout || ox >= W) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (oy + i >= H) continue;
            const float s12 = u[i][1] + u[i][2], d12 = u[i][1] - u[i][2], s34 = u[i][3] + u[i][4], d34 = u[i][3] - u[i][4];
            float4 v = make_float4((u[i][0] + s12) + s34, fmaf(2.f, d34, d12), fmaf(4.f, s34, s12), fmaf(8.f, d34, d12) + u[i][5]);
            v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            const size_t o0 = (size_t)co * HWo + (size_t)(oy + i) * W + ox;
            if (vec_ok) {      // W % 4 == 0 and ox % 4 == 0: the four pixels are inside and 16-B aligned
                if (rin) { const float4 q = *reinterpret_cast<const float4*>(rin + o0); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
                if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(yout + o0) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (ox + k >= W) break;
                    float t = vv[k];
                    if (rin) t += rin[o0 + k];
                    if (relu) t = fmaxf(t, 0.f);
                    yout[o0 + k] = t;
                }
            }
        }
    }
}

}  // namespace heal

using namespace heal;

// u_frag: U = G g G^T of the [Cout,Cin,3,3] filter bank in the lane-major order the kernel reads (ops.conv3x3_winograd4_fragments):
//   [ceil(Cout/32)][ceil(Cin/16)][wave 8][lane 64][xi_i 9][ks 4] floats,
//   value = U[xi = 9 (wave & 3) + xi_i][co = 32 mb + 16 (wave >> 2) + (lane & 15)][ci = 16 chunk + 4 ks + (lane >> 4)], zero-padded.
extern "C" int heal_conv3x3_winograd4(const float* x, const float* u_frag, const float* bias, const float* residual, int n, int cin,
                                      int cout, int H, int W, int relu, float* y, void* stream) {
    HEAL_REQUIRE(n >= 1 && H >= 1 && W >= 1 && cin >= 1 && cout >= 1, "conv3x3_winograd4: bad shape");
    HEAL_REQUIRE(x && u_frag && y, "conv3x3_winograd4: null pointer");
    HEAL_REQUIRE(((uintptr_t)u_frag & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!residual || ((uintptr_t)residual & 15) == 0),
                 "conv3x3_winograd4: 16-B alignment");
    const int nchunks = (cin + W4_KC - 1) / W4_KC, mblocks = (cout + 31) / 32;
    const int tiles_x = ceil_div(W, W4_TC), tiles_y = ceil_div(H, W4_TR);
    HEAL_REQUIRE((long long)tiles_x * tiles_y <= 2147483647ll / 4 && n <= 65535 && mblocks <= 65535,
                 "conv3x3_winograd4: map too large for the launch grid");
    const dim3 grid(tiles_x * tiles_y, n, mblocks);
    k_conv3x3_wino4<<<grid, 512, 0, (hipStream_t)stream>>>(x, reinterpret_cast<const float4*>(u_frag), bias, residual, cin, nchunks,
                                                          cout, H, W, tiles_x, relu, y);
    HEAL_LAUNCH_CHECK();
    return 0;
}
