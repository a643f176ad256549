// Pointwise (1x1, stride 1) convolution with the epilogue fused: y = act(W x + b (+ residual)), NCHW fp32.
// Replaces the `conv1x1 -> BatchNorm -> (+identity) -> ReLU` sequences of the ResNeXt bottlenecks
// (opencood/models/sub_modules/resblock.py:95-121) that the library path runs as a Tensile GEMM plus a separate
// bias/residual/ReLU pass over the output.
//
// Per image this is the GEMM  Y[Cout, HW] = W[Cout, Cin] · X[Cin, HW]  with X and Y already in the layout MFMA
// wants for B and D (pixels = columns), so nothing is transposed:
//   * block = 4 waves, output tile BM channels x 128 pixels; wave w owns BM/4 rows (BM/64 m-tiles) x 8 n-tiles,
//     accumulators in registers (v_mfma_f32_16x16x4_f32; fp32 in, fp32 accumulate);
//   * B (activations): 32-channel x 128-pixel chunks staged through LDS with 512-B coalesced rows (16 B/lane),
//     double-buffered against the MFMA loop; row stride 144 floats -> the four k-rows of a fragment read fall in
//     disjoint bank groups (conflict-free ds_read_b32);
//   * A (weights): pre-laid on the host in fragment order frag[mt][ks / 4][lane][ks % 4] (ops.conv1x1_fragments), so a wave's
//     A operands of FOUR k-steps are one coalesced 1-KiB load (16 B per lane) straight from L2 -- no LDS, no shuffles;
//   * epilogue in registers: + bias[co] (+ residual) -> ReLU | SiLU -> store (64-B segments per 16-pixel run);
//   * XCD-contiguous block order with the Cout-chunk index fastest: the (<=4) blocks that re-read one pixel tile
//     for different output-channel chunks share an L2.
// Roofline: HBM-bound for Cin,Cout <= 128 (ridge 20 FLOP/B in fp32), MFMA-bound above; §6 of DESIGN.md.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BM, int BN, int KC, int STRIDE>
__global__ __launch_bounds__(256) void k_conv1x1(const float* __restrict__ x, const float* __restrict__ wfrag,
                                                const float* __restrict__ bias, const float* __restrict__ res,
                                                const float* __restrict__ in_scale, int Cin, int Kpad, int Cout,
                                                int HW, int Wo, int in_W, int in_HW, int act, int out_pm,
                                                int d2s_k, int d2s_ctot, int d2s_coff, int n_img, int ksplit,
                                                float* __restrict__ y) {
    // ksplit > 1 (small maps with a deep reduction: 36 blocks for the 1152 -> 192 projection at 12 x 16 x 4 pixels): grid.z =
    // ksplit * n_img, block z reduces chunk range ks of image n and writes its PARTIAL sum to y[ks][n][Cout][HW] (no bias /
    // residual / activation: k_conv1x1_splitk_reduce adds the partials in split order and applies them).
    // out_pm: 0 NCHW | 1 pixel-major | 2 depth-to-space INTO A SLICE of a wider NCHW tensor: output channel co of pixel
    // (h, w) lands at channel d2s_coff + co / k^2, pixel (h k + (co % k^2) / k, w k + co % k) of y [n, d2s_ctot, H k, W k] --
    // ConvTranspose2d(kernel = stride = k) + pixel shuffle + torch.cat in the epilogue (k = 1: a plain channel-offset write).
    // HW / Wo: OUTPUT pixels per image / per row; in_W / in_HW: input row width / pixels per image.  STRIDE 1: in == out.
    constexpr int MT = BM / 64;      // m-tiles per wave
    constexpr int NT = BN / 16;      // n-tiles per wave
    constexpr int LD = BN + 16;      // LDS row stride (floats): LD % 64 == 16 -> k-rows 0..3 hit banks 0-15,16-31,...
    constexpr int KS = KC / 4;       // k-steps per chunk
    constexpr int TPR = BN / 4;      // staging: threads per row (one float4 each)
    constexpr int RPP = 256 / TPR;   // rows per pass
    constexpr int NPASS = KC / RPP;  // float4 loads per thread per chunk
    static_assert(NPASS >= 1 && KC % RPP == 0, "bad staging shape");
    __shared__ __attribute__((aligned(16))) float sB[2][KC][LD];
    const Block3 bk = xcd_block();  // x: Cout chunk, y: pixel tile, z: image
    const int m0 = bk.x * BM, p0 = bk.y * BN, n = bk.z % n_img, kpart = bk.z / n_img;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int lk = l >> 4, ln = l & 15;
    const float* __restrict__ xin = x + (size_t)n * Cin * in_HW;
    const int ksteps = Kpad / 4;    // k-steps of 4 channels over the (zero-padded) K of the fragment layout
    const int nchunks_all = Kpad / KC, cps = (nchunks_all + ksplit - 1) / ksplit;
    const int cbeg = kpart * cps, nchunks = min(cbeg + cps, nchunks_all);   // this block's chunk range [cbeg, nchunks)
    const float* __restrict__ scl = in_scale ? in_scale + (size_t)n * Cin : nullptr;
    const int sp = (threadIdx.x % TPR) * 4, sc = threadIdx.x / TPR;
    const bool pix_ok = p0 + sp < HW;  // HW % 4 == 0 (checked by the host): a float4 is all-in or all-out

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 stage[NPASS];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int ch = c * KC + sc + RPP * i;
            if (pix_ok && ch < Cin) {
                float4 v;
                if constexpr (STRIDE == 1) {
                    v = *reinterpret_cast<const float4*>(xin + (size_t)ch * HW + p0 + sp);
                } else {  // 4 consecutive output pixels of one row (Wo % 4 == 0) -> every STRIDE-th input pixel
                    const int p = p0 + sp, oy = p / Wo, ox = p - oy * Wo;
                    const float* src = xin + (size_t)ch * in_HW + (size_t)(oy * STRIDE) * in_W + ox * STRIDE;
                    v = make_float4(src[0], src[STRIDE], src[2 * STRIDE], src[3 * STRIDE]);
                }
                if (scl) { const float g = scl[ch]; v.x *= g; v.y *= g; v.z *= g; v.w *= g; }
                stage[i] = v;
            } else {
                stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) *reinterpret_cast<float4*>(&sB[buf][sc + RPP * i][sp]) = stage[i];
    };

    // A fragments of one chunk: MT x KS coalesced 256-B loads per wave (weights stay L2-resident)
    float a_cur[MT][KS], a_nxt[MT][KS];
    auto load_a = [&](int c, float (&a)[MT][KS]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mtg = (m0 >> 4) + wave * MT + mt;  // global m-tile
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {     // four k-steps per 16-B load: frag[mt][ks / 4][lane][ks % 4]
                const float4 v = *reinterpret_cast<const float4*>(wfrag + (((size_t)mtg * (ksteps / 4) + (c * KS) / 4 + q) * 64 + l) * 4);
                a[mt][4 * q] = v.x; a[mt][4 * q + 1] = v.y; a[mt][4 * q + 2] = v.z; a[mt][4 * q + 3] = v.w;
            }
        }
    };

    load_a(cbeg, a_cur);
    load_chunk(cbeg);
    store_chunk(0);
    __syncthreads();
    for (int c = cbeg; c < nchunks; ++c) {
        const int buf = (c - cbeg) & 1;
        // Software pipeline: the NEXT chunk's A fragments and B rows are requested before this chunk's MFMAs, which
        // depend on nothing outstanding (vmcnt is in-order: a load the MFMAs needed behind the HBM-latency B loads
        // would stall the matrix pipe for the whole round trip).
        if (c + 1 < nchunks) {
            load_a(c + 1, a_nxt);
            load_chunk(c + 1);
        }
        // B fragments of k-step ks+1 are read from LDS while the MFMAs of k-step ks run (register double buffer)
        // Column j of n-tile nt is PIXEL (nt / 4) * 64 + 4 j + nt % 4 of the block's tile (columns of a GEMM can be numbered freely):
        // a lane's B operands of four n-tiles are four consecutive pixels = ONE ds_read_b128 (16 lanes x 16 B = a 256-B run per
        // k-row, the four k-rows 16 banks apart), and its results are four consecutive pixels of a channel = one 16-B store.
        float bfr[2][NT];
        auto read_b = [&](int ks, float (&dst)[NT]) {
#pragma unroll
            for (int gq = 0; gq < NT / 4; ++gq) {
                const float4 v = *reinterpret_cast<const float4*>(&sB[buf][ks * 4 + lk][gq * 64 + 4 * ln]);
                dst[4 * gq] = v.x; dst[4 * gq + 1] = v.y; dst[4 * gq + 2] = v.z; dst[4 * gq + 3] = v.w;
            }
        };
        read_b(0, bfr[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) read_b(ks + 1, bfr[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE this k-step's MFMAs (the scheduler sinks it)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt][ks], bfr[ks & 1][nt], acc[mt][nt], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_cur[mt][ks] = a_nxt[mt][ks];
    }

    if (out_pm == 1) {
        // Pixel-major output y[n][pixel][Cout] (what K4's lift + BEV pool reads: one pixel's channels are one contiguous
        // row).  MFMA leaves D[row = lk*4 + r][col = ln]: a lane holds 4 CONSECUTIVE channels of one pixel -> one 16-B
        // store per (mt, nt), the four lk groups complete a 64-B run per pixel.  No residual in this layout.
        float* __restrict__ yout = y + (size_t)n * HW * Cout;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int co = m0 + (wave * MT + mt) * 16 + lk * 4;
            if (co >= Cout) continue;  // Cout % 4 == 0 (host): a float4 is all-in or all-out
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + co);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int p = p0 + (nt / 4) * 64 + 4 * ln + (nt & 3);
                if (p >= HW) continue;
                float4 v = make_float4(acc[mt][nt][0] + bv.x, acc[mt][nt][1] + bv.y, acc[mt][nt][2] + bv.z,
                                       acc[mt][nt][3] + bv.w);
                if (act == 1) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (act == 2) {
                    v.x = v.x / (1.f + expf(-v.x)); v.y = v.y / (1.f + expf(-v.y));
                    v.z = v.z / (1.f + expf(-v.z)); v.w = v.w / (1.f + expf(-v.w));
                }
                *reinterpret_cast<float4*>(yout + (size_t)p * Cout + co) = v;
            }
        }
        return;
    }
    // Epilogue.  MFMA leaves D[row = lk*4 + r][col = ln] per (mt, nt); with the interleaved column numbering a lane holds, per channel
    // row, FOUR CONSECUTIVE PIXELS in the tiles 4 gq .. 4 gq + 3: bias / residual / activation / store work on 16-B pieces straight
    // from the accumulators (rounds 1-3 transposed every tile through LDS for that), 256-B runs per channel and instruction.
    float* __restrict__ yout = y + ((size_t)kpart * n_img + n) * Cout * HW;
    const float* __restrict__ rin = res ? res + (size_t)n * Cout * HW : nullptr;
    constexpr int NG = NT / 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // residual pieces of this m-tile first (unconditional loads on clamped addresses), all in flight together
        float4 rq[4][NG];
        if (rin) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) {
                    const int co = min(m0 + (wave * MT + mt) * 16 + lk * 4 + r, Cout - 1), p = min(p0 + gq * 64 + 4 * ln, HW - 4);
                    rq[r][gq] = *reinterpret_cast<const float4*>(rin + (size_t)co * HW + p);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = m0 + (wave * MT + mt) * 16 + lk * 4 + r;
            if (co >= Cout) continue;
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const int p = p0 + gq * 64 + 4 * ln;
                if (p >= HW) continue;                                  // HW % 4 == 0 (host): a piece is all-in or all-out
                float4 v = make_float4(acc[mt][4 * gq][r] + bv, acc[mt][4 * gq + 1][r] + bv, acc[mt][4 * gq + 2][r] + bv,
                                       acc[mt][4 * gq + 3][r] + bv);
                const size_t o = (size_t)co * HW + p;
                if (rin) { const float4 q = rq[r][gq]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
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
                if (out_pm != 2) {
                    *reinterpret_cast<float4*>(yout + o) = v;
                } else {
                    const int kk = d2s_k * d2s_k, Ho = HW / Wo;
                    const int c = co / kk, rr = co - c * kk, dy = rr / d2s_k, dx = rr - dy * d2s_k;
                    const int hh = p / Wo, ww = p - hh * Wo;      // Wo % 4 == 0 (host): the four pixels share a row
                    float* dst = y + (((size_t)n * d2s_ctot + d2s_coff + c) * ((size_t)Ho * d2s_k) + (size_t)hh * d2s_k + dy) *
                                         ((size_t)Wo * d2s_k) + (size_t)ww * d2s_k + dx;
                    if (d2s_k == 1) {
                        *reinterpret_cast<float4*>(dst) = v;
                    } else {
                        dst[0] = v.x; dst[d2s_k] = v.y; dst[2 * d2s_k] = v.z; dst[3 * d2s_k] = v.w;
                    }
                }
            }
        }
    }
}


// out[n][co][p] = act(sum_ks part[ks][n][co][p] + bias[co] (+ residual)), the partials added in split order (deterministic)
__global__ __launch_bounds__(256) void k_conv1x1_splitk_reduce(const float4* __restrict__ part, const float* __restrict__ bias,
                                                              const float4* __restrict__ res, int ksplit, int Cout, int HW4,
                                                              size_t total4, int act, float4* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 v = part[i];
    for (int k = 1; k < ksplit; ++k) {
        const float4 q = part[(size_t)k * total4 + i];
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    const float bv = bias ? bias[(i / HW4) % Cout] : 0.f;
    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
    if (res) { const float4 q = res[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (act == 1) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (act == 2) {
        v.x = v.x / (1.f + expf(-v.x)); v.y = v.y / (1.f + expf(-v.y)); v.z = v.z / (1.f + expf(-v.z)); v.w = v.w / (1.f + expf(-v.w));
    } else if (act == 3) {
        v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752f)); v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752f));
        v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752f)); v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752f));
    }
    y[i] = v;
}

}  // namespace heal

using namespace heal;

namespace heal {
// shared with the Winograd split-K path (conv3x3.hip)
int splitk_reduce_launch(const float* partials, const float* bias, const float* residual, int ksplit, int n, int cout, int HW,
                         int act, float* y, hipStream_t s, hipEvent_t ev_stop) {
    const size_t total4 = (size_t)n * cout * HW / 4;
    HEAL_LAUNCH_EV2(k_conv1x1_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, (hipEvent_t) nullptr, ev_stop,
                    reinterpret_cast<const float4*>(partials), bias, reinterpret_cast<const float4*>(residual), ksplit, cout,
                    HW / 4, total4, act, reinterpret_cast<float4*>(y));
    HEAL_LAUNCH_CHECK();
    return 0;
}
}  // namespace heal

static int conv1x1_launch(const float* x, const float* weight_frag, const float* bias, const float* residual,
                          const float* in_scale, int n, int cin, int cout, int H, int W, int stride, int act,
                          int out_pixel_major, int d2s_k, int d2s_ctot, int d2s_coff, float* y, void* stream,
                          int ksplit = 1, float* partials = nullptr) {
    HEAL_REQUIRE(n >= 1 && H >= 1 && W >= 1 && cin >= 1 && cout >= 1, "conv1x1: bad shape");
    HEAL_REQUIRE(stride == 1 || stride == 2, "conv1x1: stride must be 1 or 2 (got %d)", stride);
    const int kpad = (cin + 31) / 32 * 32, mpad = (cout + 63) / 64 * 64;  // dims of the zero-padded fragment layout
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1, HW = Ho * Wo;
    if (stride == 1) HEAL_REQUIRE(HW % 4 == 0, "conv1x1: H*W must be a multiple of 4 (got %d)", HW);
    else HEAL_REQUIRE(Wo % 4 == 0, "conv1x1: output width must be a multiple of 4 for stride 2 (got %d)", Wo);
    HEAL_REQUIRE(act >= 0 && act <= 3, "conv1x1: act must be 0 (none), 1 (ReLU), 2 (SiLU) or 3 (GELU)");
    HEAL_REQUIRE(x && weight_frag && y, "conv1x1: null pointer");
    if (out_pixel_major == 1) {
        HEAL_REQUIRE(residual == nullptr && act != 3, "conv1x1: pixel-major output takes no residual / GELU");
        HEAL_REQUIRE(cout % 4 == 0 && ((uintptr_t)bias & 15) == 0, "conv1x1: pixel-major output needs Cout %% 4 == 0 and a 16-B aligned bias");
    }
    hipStream_t s = (hipStream_t)stream;
    // Tile choice (BM, BN, KC).  Measured on MI355X at the PyramidFusion shapes (scripts/conv1x1_bench.py --sweep): the
    // smallest tile wins everywhere (64 channels x 64 pixels: 48 VGPR + 16 AGPR, 20 KB LDS -> 8 waves/SIMD); the kernel
    // is latency-bound between chunks and more resident blocks hide it better than a fatter tile's operand reuse.
    // HEAL_C1_CFG="bm,bn,kc" overrides for tuning (stride 1).
    int bm = 64, bn = 64, kc = 32;
    if (const char* e = getenv("HEAL_C1_CFG")) {
        int a_ = 0, b_ = 0, c_ = 0;
        if (stride == 1 && sscanf(e, "%d,%d,%d", &a_, &b_, &c_) == 3) { bm = a_; bn = b_; kc = c_; }
    }
    HEAL_REQUIRE(mpad % bm == 0, "conv1x1: padded Cout %d not a multiple of the tile height %d", mpad, bm);
#define HEAL_C1(BM_, BN_, KC_, ST_)                                                                              \
    if (bm == BM_ && bn == BN_ && kc == KC_ && stride == ST_) {                                                  \
        if (ksplit > 1)                                                                                          \
            HEAL_LAUNCH_EV2((k_conv1x1<BM_, BN_, KC_, ST_>), dim3(mpad / BM_, ceil_div(HW, BN_), n * ksplit), dim3(256), 0, s, \
                            ev.start, (hipEvent_t) nullptr,                                                      \
                            x, weight_frag, (const float*)nullptr, (const float*)nullptr, in_scale, cin, kpad, cout, HW, Wo, W, \
                            H * W, 0, 0, 1, 0, 0, n, ksplit, partials);                                          \
        else                                                                                                     \
            HEAL_LAUNCH_EV2((k_conv1x1<BM_, BN_, KC_, ST_>), dim3(mpad / BM_, ceil_div(HW, BN_), n), dim3(256), 0, s, \
                            ev.start, ev.stop,                                                                   \
                            x, weight_frag, bias, residual, in_scale, cin, kpad, cout, HW, Wo, W, H * W, act, out_pixel_major, \
                            d2s_k, d2s_ctot, d2s_coff, n, 1, y);                                                 \
        launched = true;                                                                                         \
    }
    bool launched = false;
    const LaunchEvents ev = take_launch_events();   // measurement hook (heal_next_launch_events): the kernel's own begin / end
    HEAL_C1(64, 64, 32, 1) HEAL_C1(64, 64, 32, 2)
    HEAL_C1(128, 128, 32, 1) HEAL_C1(64, 128, 32, 1) HEAL_C1(128, 64, 32, 1)
#undef HEAL_C1
    HEAL_REQUIRE(launched, "conv1x1: no kernel for tile (%d,%d,%d) stride %d", bm, bn, kc, stride);
    if (ksplit > 1) {
        const size_t total4 = (size_t)n * cout * HW / 4;
        HEAL_LAUNCH_EV2(k_conv1x1_splitk_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, (hipEvent_t) nullptr, ev.stop,
                        reinterpret_cast<const float4*>(partials), bias, reinterpret_cast<const float4*>(residual), ksplit, cout,
                        HW / 4, total4, act, reinterpret_cast<float4*>(y));
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_conv1x1(const float* x, const float* weight_frag, const float* bias, const float* residual,
                            const float* in_scale, int n, int cin, int cout, int H, int W, int stride, int act,
                            int out_pixel_major, float* y, void* stream) {
    HEAL_REQUIRE(out_pixel_major == 0 || out_pixel_major == 1, "conv1x1: out_pixel_major must be 0 or 1");
    return conv1x1_launch(x, weight_frag, bias, residual, in_scale, n, cin, cout, H, W, stride, act, out_pixel_major, 1, 0, 0,
                          y, stream);
}

extern "C" int heal_conv1x1_d2s(const float* x, const float* weight_frag, const float* bias, int n, int cin, int cout, int H,
                                int W, int act, int k, int dst_channels, int dst_channel_offset, float* y, void* stream) {
    HEAL_REQUIRE(k >= 1 && k <= 8 && cout % (k * k) == 0, "conv1x1_d2s: Cout must be a multiple of k^2 (k = %d)", k);
    HEAL_REQUIRE(W % 4 == 0, "conv1x1_d2s: W must be a multiple of 4 (got %d)", W);
    HEAL_REQUIRE(dst_channel_offset >= 0 && dst_channel_offset + cout / (k * k) <= dst_channels,
                 "conv1x1_d2s: channel slice [%d, %d) outside the %d destination channels", dst_channel_offset,
                 dst_channel_offset + cout / (k * k), dst_channels);
    HEAL_REQUIRE(((uintptr_t)y & 15) == 0, "conv1x1_d2s: destination must be 16-B aligned");
    return conv1x1_launch(x, weight_frag, bias, nullptr, nullptr, n, cin, cout, H, W, 1, act, 2, k, dst_channels,
                          dst_channel_offset, y, stream);
}

extern "C" size_t heal_conv1x1_splitk_workspace(int n, int cout, int H, int W, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * n * cout * H * W * sizeof(float) : 0;
}

extern "C" int heal_conv1x1_splitk(const float* x, const float* weight_frag, const float* bias, const float* residual,
                                   const float* in_scale, int n, int cin, int cout, int H, int W, int act, int ksplit, float* y,
                                   void* ws, size_t ws_bytes, void* stream) {
    const int nchunks = (cin + 31) / 32;
    HEAL_REQUIRE(ksplit >= 2 && ksplit <= nchunks, "conv1x1_splitk: ksplit must be in [2, %d] (got %d)", nchunks, ksplit);
    HEAL_REQUIRE((ksplit - 1) * ceil_div(nchunks, ksplit) < nchunks,
                 "conv1x1_splitk: %d splits of %d chunks leave an empty split (use ceil(chunks / ceil(chunks / ksplit)))", ksplit,
                 nchunks);
    HEAL_REQUIRE((H * W) % 4 == 0, "conv1x1_splitk: H*W must be a multiple of 4");
    HEAL_REQUIRE((long long)n * ksplit <= 65535, "conv1x1_splitk: n * ksplit exceeds the grid limit");
    HEAL_REQUIRE(ws && ws_bytes >= heal_conv1x1_splitk_workspace(n, cout, H, W, ksplit) && ((uintptr_t)ws & 15) == 0,
                 "conv1x1_splitk: workspace too small or misaligned");
    HEAL_REQUIRE(act >= 0 && act <= 3, "conv1x1_splitk: act must be 0..3");
    return conv1x1_launch(x, weight_frag, bias, residual, in_scale, n, cin, cout, H, W, 1, act, 0, 1, 0, 0, y, stream, ksplit,
                          (float*)ws);
}
