// Fused window attention of the V2X-ViT pyramid (opencood/models/sub_modules/mswin.py:46-80,
// BaseWindowAttention): for every (agent, ws x ws window, head)
//     O = softmax(scale * Q K^T + pos_bias) V,      Q, K, V [T = ws*ws, d]
// straight from the packed to_qkv output [L,H,W,3*m*d] to [L,H,W,m*d].  The library path materialises the window
// re-layout of q/k/v, the [B,T,T] score tensor (0.5 GB at ws = 16), its softmax and the re-layout of the result:
// ~2.5 GB of HBM traffic per attention against 0.54 GB compulsory.
//
// One block per (window, head, agent), up to 4 waves; a wave owns 16-query-row slabs.
//   S = Q K^T on v_mfma_f32_16x16x4_f32: A = Q rows straight from global (16-B runs per token), B = K^T from an LDS copy
//       of the window's K (row stride d+4: conflict-free fragment reads); the slab's whole score row block (16 x T) stays
//       in registers (T/16 accumulator tiles);
//   softmax in registers: rows live on 16-lane groups -> xor-shuffles inside the group, exp, normalise;
//   P goes through the wave's LDS slice (MFMA D layout -> A layout), O = P V with V read from global in B-fragment order
//       (64-B runs, L2-resident), result written as 64-B runs per token.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// waves per block: one per 16-row slab up to 4; the 256-token window runs 8 waves that share K AND V in LDS (150 KB, one
// block per CU, two waves per SIMD) and moves P from the MFMA D layout to the A layout with lane shuffles instead of an
// LDS slice -- no global load is left inside its MFMA loops.
template <int WS>
constexpr int wattn_waves() { return WS * WS == 256 ? 8 : ((WS * WS / 16) < 4 ? (WS * WS / 16) : 4); }

template <int WS, int D>
__global__ __launch_bounds__(64 * wattn_waves<WS>()) void k_window_attn(
    const float* __restrict__ qkv /*[L,H,W,3,m,D]*/, const float* __restrict__ bias /*[T,T] or null*/, int H, int W,
    int m, float scale, float* __restrict__ out /*[L,H,W,m*D]*/) {
    constexpr int T = WS * WS, KC = D / 4, NCB = T / 16, NB = D / 16, NW = wattn_waves<WS>();
    constexpr bool BIG = T == 256;                      // V in LDS, P through shuffles
    constexpr bool VL = T >= 64;                        // V staged in LDS next to K (round 4: also the 8 x 8 window -- its P V loop read V
                                                        // from global memory inside the MFMA loop, four exposed L2 round trips per wave)
    constexpr int KSTR = D + 4, PSTR = T + 4, VSTR = D + 16;  // VSTR % 64 == 16: k-rows of a B fragment in disjoint banks
    __shared__ __attribute__((aligned(16))) float sK[T * KSTR];
    __shared__ __attribute__((aligned(16))) float sV[VL ? T * VSTR : 4];
    __shared__ float sP[BIG ? 1 : NW][BIG ? 4 : 16 * PSTR];
    const int nww = W / WS;
    const int ih = blockIdx.x / nww, iw = blockIdx.x - ih * nww, h = blockIdx.y, l = blockIdx.z;
    const int MD = m * D;
    const size_t C3 = (size_t)3 * MD;
    const float* base = qkv + ((size_t)l * H * W) * C3 + (size_t)h * D;
    // token t of this window -> offset of its (q) row
    auto tok = [&](int t) -> size_t {
        const int y = ih * WS + t / WS, x = iw * WS + t % WS;
        return ((size_t)y * W + x) * C3;
    };
    // the first slab's Q rows are requested together with K / V (one exposed round trip instead of two)
    const int wave0 = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
    float a_first[KC];
    {
        const float* qp = base + tok(min(wave0, T / 16 - 1) * 16 + (lane0 & 15));
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) a_first[kc] = qp[kc * 4 + (lane0 >> 4)];
    }
    // stage K (chunk 1 of the packed projection)
    for (int e = threadIdx.x; e < T * (D / 4); e += 64 * NW) {
        const int t = e / (D / 4), c4 = e - t * (D / 4);
        const float4 v = *reinterpret_cast<const float4*>(base + tok(t) + MD + c4 * 4);
        *reinterpret_cast<float4*>(&sK[t * KSTR + c4 * 4]) = v;
        if constexpr (VL)
            *reinterpret_cast<float4*>(&sV[t * VSTR + c4 * 4]) =
                *reinterpret_cast<const float4*>(base + tok(t) + 2 * MD + c4 * 4);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lk = lane >> 4, ln = lane & 15;
    float* sp = sP[BIG ? 0 : wave];
    for (int slab = wave; slab < T / 16; slab += NW) {
        // ---- S = Q K^T ------------------------------------------------------------------------------------------
        float a[KC];
        if (slab == wave) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) a[kc] = a_first[kc];
        } else {
            const float* qp = base + tok(slab * 16 + ln);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) a[kc] = qp[kc * 4 + lk];
        }
        f32x4 s[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            s[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc], sK[(cb * 16 + ln) * KSTR + kc * 4 + lk], s[cb], 0, 0, 0);
        }
        // ---- scale, bias, softmax over the T keys of each row (D layout: row = lk*4 + r, col = cb*16 + ln) ----------
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[cb][r] * scale;
                if (bias) v += bias[(size_t)(slab * 16 + lk * 4 + r) * T + cb * 16 + ln];
                s[cb][r] = v;
                mx[r] = fmaxf(mx[r], v);
            }
        float sum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
            sum[r] = 0.f;
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(s[cb][r] - mx[r]);
                s[cb][r] = e;
                sum[r] += e;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum[r] += __shfl_xor(sum[r], o, 64);
            sum[r] = 1.f / sum[r];
        }
        f32x4 o[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) o[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (!BIG) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) sp[(lk * 4 + r) * PSTR + cb * 16 + ln] = s[cb][r] * sum[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- O = P V: A = P from the wave's LDS slice, B = V from global (64-B runs, L2-resident) --------------
#pragma unroll 4
            for (int kc = 0; kc < T / 4; ++kc) {
                const float p = sp[ln * PSTR + kc * 4 + lk];
                if constexpr (VL) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, sV[(kc * 4 + lk) * VSTR + nb * 16 + ln], o[nb], 0, 0, 0);
                } else {
                    const float* vp = base + tok(kc * 4 + lk) + 2 * MD;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, vp[nb * 16 + ln], o[nb], 0, 0, 0);
                }
            }
        } else {
            // ---- O = P V: P moves D layout -> A layout by shuffles.  A-lane (m = ln, k = lk) of k-step kc needs
            // P[row ln][col 4kc + lk], which sits in lane ((ln >> 2) << 4 | (4 (kc & 3) + lk)), register s[kc >> 2][ln & 3].
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[cb][r] *= sum[r];
            const int rsel = ln & 3;
#pragma unroll
            for (int kc = 0; kc < T / 4; ++kc) {
                const int srcl = ((ln >> 2) << 4) | (4 * (kc & 3) + lk);
                const float p0 = __shfl(s[kc >> 2][0], srcl, 64), p1 = __shfl(s[kc >> 2][1], srcl, 64);
                const float p2 = __shfl(s[kc >> 2][2], srcl, 64), p3 = __shfl(s[kc >> 2][3], srcl, 64);
                const float p = rsel == 0 ? p0 : rsel == 1 ? p1 : rsel == 2 ? p2 : p3;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, sV[(kc * 4 + lk) * VSTR + nb * 16 + ln], o[nb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = slab * 16 + lk * 4 + r;
            const int y = ih * WS + t / WS, x = iw * WS + t % WS;
            float* op = out + (((size_t)l * H + y) * W + x) * MD + (size_t)h * D;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) op[nb * 16 + ln] = o[nb][r];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // before the next slab overwrites the P slice
    }
}

}  // namespace heal

using namespace heal;

extern "C" int heal_window_attention(const float* qkv, const float* pos_bias, int n_agents, int H, int W, int heads,
                                     int dim_head, int window, float scale, float* out, void* stream) {
    HEAL_REQUIRE(n_agents >= 1 && heads >= 1 && H >= 1 && W >= 1, "window_attention: bad shape");
    HEAL_REQUIRE(H % window == 0 && W % window == 0, "window_attention: H, W must be multiples of the window size");
    HEAL_REQUIRE(heads <= 65535 && n_agents <= 65535, "window_attention: too many heads / agents for the grid");
    HEAL_REQUIRE(qkv && out, "window_attention: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((H / window) * (W / window), heads, n_agents);
#define HEAL_WA(WS_, D_)                                                                                        \
    if (window == WS_ && dim_head == D_) {                                                                      \
        constexpr int NW_ = wattn_waves<WS_>();                                                                  \
        k_window_attn<WS_, D_><<<grid, 64 * NW_, 0, s>>>(qkv, pos_bias, H, W, heads, scale, out);               \
        HEAL_LAUNCH_CHECK();                                                                                    \
        return 0;                                                                                               \
    }
    HEAL_WA(4, 16) HEAL_WA(8, 32) HEAL_WA(16, 64) HEAL_WA(4, 32) HEAL_WA(8, 16) HEAL_WA(8, 64) HEAL_WA(4, 64)
#undef HEAL_WA
    return set_error("window_attention: window %d with dim_head %d is not instantiated", window, dim_head);
}
