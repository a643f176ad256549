// heal_conv1x1_split (round 6, OPT-IN: HEAL_ARITH=bf16x6 | bf16x9 -- never the default, never the benchmark's headline).
//
// The pointwise convolution Y[Cout, HW] = W[Cout, Cin] . X[Cin, HW] of heal_conv1x1 with fp32 inputs and fp32 accumulation, evaluated on the
// BF16 matrix cores by operand splitting.  The fp32 MFMA runs at 1/16 of the bf16 MFMA rate (157 vs 2 500 TFLOP/s), and every dense
// kernel of the path sits at 0.5 - 0.65 of it: the arithmetic, not the schedule, bounds the step.  An fp32 number is the
// sum of three bf16 numbers (24 = 8 + 8 + 8 significand bits):  a = a_h + a_m + a_l  with  a_h = bf16(a), a_m = bf16(a - a_h),
// a_l = bf16(a - a_h - a_m)  (round to nearest; the residuals are exact in fp32), so
//       a b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h) + [a_m b_l + a_l b_m + a_l b_l]
// and every partial product of two bf16 values is EXACT in fp32 (16 significand bits).  NPROD = 6 drops the bracket (terms below
// 2^-25 |a b|: the size of the rounding error of ONE fp32 product), NPROD = 9 keeps it (the products are then exact: the result differs
// from the fp32-MFMA kernel only in the order of the fp32 accumulation).  Six bf16 MFMAs cost 6 / 16 of one fp32 MFMA of the same shape.
//
//   block  = 128 output channels x 128 pixels of one image, 4 waves x (64 x 64) = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16;
//   A      = the weights, split into three bf16 planes and pre-laid in fragment order by the host (ops.conv1x1_split_fragments):
//            a lane's 8 consecutive k of a row are one 16-B load from L2, no LDS;
//   B      = the activations: a chunk of 32 input channels x 128 pixels is read as fp32 (thread = one pixel x two groups of 8 channels,
//            lanes along the pixels: coalesced), split in registers (v_cvt_pk_bf16_f32) and stored as three bf16 planes [pixel][k]
//            (80-B rows: conflict-free 16-B fragment reads), double-buffered;
//   acc    = two fp32 accumulators per output block: the leading products a_h b_h and everything else (the corrections are ~2^-8 of
//            the sum: accumulating them apart keeps their own rounding out of the result), added in the epilogue with bias / residual / act.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16s;

constexpr int CS_BM = 128, CS_BK = 32;
constexpr int CS_ROWB = 80;                       // bytes per pixel row of a B plane (32 bf16 = 64 B + 16 B pad)

struct Split3 { bf16x8 h, m, l; };

__device__ __forceinline__ Split3 split8(const float* v) {
    Split3 s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r1 = v[i] - (float)h;         // exact: v and h share the leading 8 significand bits
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;           // exact
        s.h[i] = h; s.m[i] = m; s.l[i] = (__bf16)r2;
    }
    return s;
}

// Epilogue of a wave's 64 x 64 tile: the 32x32 D blocks go through a wave-private LDS slice [32 channels][64 pixels] so that bias / residual /
// activation and the stores work on 16-B pieces (128-B runs per channel).  The first version stored one dword per lane and register -- 64
// scalar stores per lane: the anatomy (HEAL_SPLIT_DBG=15: no loads, no splitting, no MFMAs) still took 32 of the kernel's 64 us.
template <int NA, int NB>                          // row blocks / pixel blocks (32 x 32 each) of a wave's tile; slice row stride 32 NB + 4 floats
__device__ __forceinline__ void split_epilogue(float* __restrict__ se /* this wave's [32][32 NB + 4] floats */, const f32x16s (&acc)[NA][NB],
                                               const f32x16s (&cor)[NA][NB], const float* __restrict__ bias,
                                               const float* __restrict__ residual, float* __restrict__ y, int img, int cout, int HW,
                                               int co0 /* first channel of the wave tile */, int px_base /* first pixel */, int act) {
    constexpr int CS_ES = 32 * NB + 4, P4 = 8 * NB;    // float4 pieces per slice row
    const int l = threadIdx.x & 63, lj = l & 31, kb = l >> 5;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                se[((r & 3) + 8 * (r >> 2) + 4 * kb) * CS_ES + 32 * b + lj] = acc[a][b][r] + cor[a][b][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) {
            const int idx = l + 64 * i, cl = idx / P4, p4 = idx % P4;
            const int co = co0 + 32 * a + cl, px = px_base + 4 * p4;
            if (px >= HW) continue;                // HW % 4 == 0 (host): a 16-B piece is inside or outside
            float4 v = *reinterpret_cast<const float4*>(&se[cl * CS_ES + 4 * p4]);
            const size_t o = ((size_t)img * cout + co) * HW + px;
            if (bias) { const float bv = bias[co]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
            if (residual) { const float4 rv = *reinterpret_cast<const float4*>(residual + o); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
            float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (act == 1) t[k] = fmaxf(t[k], 0.f);
                else if (act == 2) t[k] = t[k] / (1.f + __expf(-t[k]));
                else if (act == 3) t[k] = 0.5f * t[k] * (1.f + erff(t[k] * 0.70710678118654752f));
            }
            *reinterpret_cast<float4*>(y + o) = make_float4(t[0], t[1], t[2], t[3]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NPROD, int NA, int NB>      // NA row blocks x NB pixel blocks per wave: the block is 64 NA channels x 64 NB pixels
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void k_conv1x1_split(const float* __restrict__ x, const uint4* __restrict__ wfrag,
                                                         const float* __restrict__ bias, const float* __restrict__ residual,
                                                         int cin, int cout, int HW, int act, float* __restrict__ y, int dbg) {
    // dbg (HEAL_SPLIT_DBG, timing anatomy only): 1 no MFMAs, 2 no splitting (hi plane only computed), 4 no activation loads, 8 no weight loads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1, lj = l & 31, kb = l >> 5;
    constexpr int BN = 64 * NB, PLANE = BN * CS_ROWB, BUF = 3 * PLANE, NS = 8 * NB;   // NS staged values per thread and chunk
    const int px0 = blockIdx.x * BN, img = blockIdx.z;
    const int co0 = blockIdx.y * 64 * NA;         // first output channel of the block; the fragments are laid out per 128 channels
    const int ct = co0 >> 7, rb0 = (co0 & 127) >> 5;
    const int nchunks = cin / CS_BK;
    const float* __restrict__ xi = x + (size_t)img * cin * HW;

    // staging role: pixel spx, NB groups of 8 channels of the chunk starting at channel NS sh
    const int spx = tid & (BN - 1), sh = tid / BN;
    const int gpx = min(px0 + spx, HW - 1);       // clamped: loads are unconditional, stores masked
    float stage[NS];
    auto load = [&](int c) {
        if (dbg & 4) return;
        const float* p = xi + (size_t)(c * CS_BK + NS * sh) * HW + gpx;
#pragma unroll
        for (int i = 0; i < NS; ++i) stage[i] = p[(size_t)i * HW];
    };
    auto store = [&](int buf) {
        unsigned char* base = smem + buf * BUF + spx * CS_ROWB + 2 * NS * sh;   // k = NS sh .. -> byte offset 2 NS sh
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            Split3 s;
            if (dbg & 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s.h[i] = (__bf16)stage[8 * g + i]; s.m[i] = s.h[i]; s.l[i] = s.h[i]; }
            } else s = split8(stage + 8 * g);
            *reinterpret_cast<bf16x8*>(base + 16 * g) = s.h;
            *reinterpret_cast<bf16x8*>(base + PLANE + 16 * g) = s.m;
            *reinterpret_cast<bf16x8*>(base + 2 * PLANE + 16 * g) = s.l;
        }
    };

    f32x16s acc[NA][NB], cor[NA][NB];             // [row block][pixel block]: leading products / corrections
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; cor[a][b][r] = 0.f; }

    // A fragments: wfrag[((ct * nchunks + c) * 2 + s) * 4 + rb][plane][lane]
    auto afrag = [&](int c, int s, int rb, int p) -> bf16x8 {
        const uint4 v = wfrag[((((size_t)(ct * nchunks + ((dbg & 8) ? 0 : c)) * 2 + ((dbg & 8) ? 0 : s)) * 4 + rb) * 3 + p) * 64 + l];
        return *reinterpret_cast<const bf16x8*>(&v);
    };
#pragma unroll
    for (int i = 0; i < NS; ++i) stage[i] = 1.f;
    load(0);
    store(0);
    bf16x8 A[NA][3], An[NA][3];                   // this k16-step's weight fragments / the next step's, requested one step ahead
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int p = 0; p < 3; ++p) { A[a][p] = afrag(0, 0, rb0 + NA * wr + a, p); An[a][p] = A[a][p]; }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load(c + 1);         // in flight during the MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the next step's A fragments (L2) are requested BEFORE this step's MFMAs: their round trip hides under 24 - 36 MFMAs
            // (loaded right in front of their use they were 12 exposed L2 latencies per chunk: the first version ran below the fp32 kernel)
            const int cn = s ? c + 1 : c, sn = s ^ 1;
            if (cn < nchunks) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int p = 0; p < 3; ++p) An[a][p] = afrag(cn, sn, rb0 + NA * wr + a, p);
            }
            bf16x8 B[NB][3];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    B[b][p] = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + p * PLANE +
                                                               (32 * NB * wc + 32 * b + lj) * CS_ROWB + (16 * s + 8 * kb) * 2);
            __builtin_amdgcn_sched_barrier(0);
            // term by term over the four output blocks (consecutive MFMAs hit different accumulators); smallest terms first
#define CS_TERM(DST, PA, PB)                                                                                        \
    _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                                  \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                              \
            DST[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][PA], B[b][PB], DST[a][b], 0, 0, 0);
            if (!(dbg & 1)) {
            if (NPROD == 9) {
                CS_TERM(cor, 2, 2) CS_TERM(cor, 1, 2) CS_TERM(cor, 2, 1)
            }
            CS_TERM(cor, 0, 2) CS_TERM(cor, 2, 0) CS_TERM(cor, 1, 1) CS_TERM(cor, 0, 1) CS_TERM(cor, 1, 0)
            CS_TERM(acc, 0, 0)
            }
#undef CS_TERM
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) A[a][p] = An[a][p];
        }
        if (c + 1 < nchunks) store(buf ^ 1);
        __syncthreads();
    }

    // epilogue through LDS (the plane buffers are free after the loop's last barrier): 16-B stores
    split_epilogue<NA, NB>(reinterpret_cast<float*>(smem) + w * 32 * (32 * NB + 4), acc, cor, bias, residual, y, img, cout, HW,
                           co0 + 32 * NA * wr, px0 + 32 * NB * wc, act);
}


// Measured and removed (round 6, scripts/split_gemm_bench.py): a PRODUCER / CONSUMER version -- blocks of eight waves, waves 0-3 only
// multiply, waves 4-7 only load (two chunks ahead), split and store into a ring of three LDS slots -- was 1.1 - 1.4x SLOWER than this kernel at
// every shape (512 -> 256 @5 x 64^2: 72.7 vs 55.8 us): one block per CU (92 KB of LDS, 8 x 212 registers) leaves the 320-tile shapes with two
// sequential rounds on a quarter of the CUs, and the producers' split arithmetic shares the SIMD's issue slots with the MFMA wave.  What DID
// matter (anatomy, HEAL_SPLIT_DBG): the epilogue -- one dword store per lane and register cost 32 of 64 us; through LDS as 16-B pieces: 55.8 us.

}  // namespace heal

using namespace heal;

extern "C" int heal_conv1x1_split_supported(int cin, int cout, int H, int W) {
    return cin >= 32 && cin % CS_BK == 0 && cout % CS_BM == 0 && (long long)H * W >= 128 && ((long long)H * W) % 4 == 0;
}

extern "C" int heal_conv1x1_split(const float* x, const void* weight_frag, const float* bias, const float* residual, int n, int cin,
                                  int cout, int H, int W, int act, int n_products, float* y, void* stream) {
    HEAL_REQUIRE(x && weight_frag && y, "conv1x1_split: null pointer");
    HEAL_REQUIRE(heal_conv1x1_split_supported(cin, cout, H, W), "conv1x1_split: needs cin %% 32 == 0, cout %% 128 == 0 (got %d -> %d)", cin, cout);
    HEAL_REQUIRE(n_products == 6 || n_products == 9, "conv1x1_split: n_products must be 6 or 9");
    HEAL_REQUIRE(((uintptr_t)weight_frag & 15) == 0, "conv1x1_split: weight fragments must be 16-B aligned");
    const int HW = H * W;
    HEAL_REQUIRE(n >= 1 && n <= 65535 && cout / CS_BM <= 65535, "conv1x1_split: grid limit");
    const uint4* wf = reinterpret_cast<const uint4*>(weight_frag);
    const int dbg = HEAL_DEBUG_ENV("HEAL_SPLIT_DBG");
    // pixels per block: 128 when that still gives every CU two or more blocks and a half, else 64 (twice the blocks, three resident per CU:
    // the 64^2 levels of the fusion pyramid give 320 blocks of 128 pixels for 256 CUs -- two rounds on a quarter of the chip)
    // block shape (channels x pixels): 128 x 64 (default) | 64 x 128 | 128 x 128.  Measured at the scene's shapes (scripts/split_gemm_bench.py,
    // bf16x6, 512 -> 256 at 5 x 64^2): 50.6 | 58.4 | 57.1 us (exact-fp32 kernel 55.6) -- the 128 x 128 block leaves the 64^2 pyramid levels with
    // 320 blocks for 256 CUs; the 64 x 128 block reads half the weight fragments per MFMA but needs 61 KB of LDS (two blocks per CU instead of
    // three) and loses; 128 x 64 wins at every shape.
    static const int shape = []() { const char* e = getenv("HEAL_SPLIT_TILE"); return e ? atoi(e) : 1; }();   // 1 128x64 | 2 64x128 | 3 128x128
    const int na = (shape == 2) ? 1 : 2, nb = (shape == 2 || shape == 3) ? 2 : 1;
    const dim3 grid(ceil_div(HW, 64 * nb), cout / (64 * na), n);
    const size_t lds = (size_t)2 * 3 * 64 * nb * CS_ROWB;      // 61 440 B (two blocks per CU) | 30 720 B
#define HEAL_CS(NP_, NA_, NB_) HEAL_LAUNCH_EV((k_conv1x1_split<NP_, NA_, NB_>), grid, dim3(256), lds, (hipStream_t)stream, x, wf, bias, residual, cin, cout, HW, act, y, dbg)
    if (n_products == 6) { if (na == 2 && nb == 2) HEAL_CS(6, 2, 2); else if (na == 2) HEAL_CS(6, 2, 1); else HEAL_CS(6, 1, 2); }
    else { if (na == 2 && nb == 2) HEAL_CS(9, 2, 2); else if (na == 2) HEAL_CS(9, 2, 1); else HEAL_CS(9, 1, 2); }
#undef HEAL_CS
    HEAL_LAUNCH_CHECK();
    return 0;
}
