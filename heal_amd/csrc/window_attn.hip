// Fused window attention of the V2X-ViT pyramid (opencood/models/sub_modules/mswin.py:46-80,
// BaseWindowAttention): for every (agent, ws x ws window, head)
//     O = softmax(scale * Q K^T + pos_bias) V,      Q, K, V [T = ws*ws, d]
// straight from the packed to_qkv output [L,H,W,3*m*d] to [L,H,W,m*d].  The library path materialises the window
// re-layout of q/k/v, the [B,T,T] score tensor (0.5 GB at ws = 16), its softmax and the re-layout of the result:
// ~2.5 GB of HBM traffic per attention against 0.54 GB compulsory.
//
// One block per (window, head, agent), up to 4 waves (8 for the 256-token window); a wave owns 16-query slabs.  Round 4: the
// TRANSPOSED formulation -- no operand ever changes layout between the two GEMMs:
//   S^T = K Q^T on v_mfma_f32_16x16x4_f32: A = K rows from an LDS copy of the window's K, B = Q rows straight from global; both
//       operands use the SAME permutation of the reduction index (lane group lk takes d = lk * D/4 + step), so a lane's operand
//       values of four consecutive steps are 16 contiguous bytes: ds_read_b128 / 16-B global loads instead of one b32 per MFMA;
//   D layout of S^T: lane (lk, ln) holds keys cb * 16 + lk * 4 + {0..3} of QUERY ln -- a query's scores sit in ONE lane column:
//       scale, relative-position bias (16-B loads), softmax with two xor-shuffles (across lk) per reduction;
//   O^T = V^T P^T: reduction step (cb, r) takes key cb * 16 + lk * 4 + r from lane group lk -- exactly register r of score tile cb:
//       the probabilities ARE the B operand (rounds 2-3 moved P through an LDS slice or 256 ds_bpermute per slab); A = V from LDS
//       (row stride D + 4: the four lane groups, four rows apart, fall in disjoint bank groups);
//   D layout of O^T: lane (lk, ln) holds channels nb * 16 + lk * 4 + {0..3} of query ln: one 16-B store per tile.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// waves per block: one per 16-query slab up to 4; the 256-token window runs 8 waves (K and V of the window in LDS: 139 KB, one
// block per CU, two waves per SIMD)
template <int WS>
constexpr int wattn_waves() { return WS * WS == 256 ? 8 : ((WS * WS / 16) < 4 ? (WS * WS / 16) : 4); }

template <int WS, int D>
__global__ __launch_bounds__(64 * wattn_waves<WS>()) void k_window_attn(
    const float* __restrict__ qkv /*[L,H,W,3,m,D]*/, const float* __restrict__ bias /*[T,T] or null*/, int H, int W,
    int m, float scale, float* __restrict__ out /*[L,H,W,m*D]*/) {
    constexpr int T = WS * WS, NCB = T / 16, NB = D / 16, NW = wattn_waves<WS>();
    constexpr int DQ = D / 4;                            // reduction indices per lane group (d = lk * DQ + step)
    constexpr int KSTR = D + 4, VSTR = D + 4;            // row strides (words): 16-B aligned; 4 VSTR = 16 (mod 64)
    __shared__ __attribute__((aligned(16))) float sK[T * KSTR];
    __shared__ __attribute__((aligned(16))) float sV[T * VSTR];
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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lk = lane >> 4, ln = lane & 15;
    // Q rows of the first slab: requested together with K / V (one exposed round trip); lane (lk, ln): query ln, d = lk * DQ ..
    float4 qf[DQ / 4];
    {
        const float* qp = base + tok(min(wave, T / 16 - 1) * 16 + ln) + lk * DQ;
#pragma unroll
        for (int i = 0; i < DQ / 4; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + 4 * i);
    }
    // stage K and V (chunks 1 and 2 of the packed projection)
    for (int e = threadIdx.x; e < T * (D / 4); e += 64 * NW) {
        const int t = e / (D / 4), c4 = e - t * (D / 4);
        const float4 kv = *reinterpret_cast<const float4*>(base + tok(t) + MD + c4 * 4);
        const float4 vv = *reinterpret_cast<const float4*>(base + tok(t) + 2 * MD + c4 * 4);
        *reinterpret_cast<float4*>(&sK[t * KSTR + c4 * 4]) = kv;
        *reinterpret_cast<float4*>(&sV[t * VSTR + c4 * 4]) = vv;
    }
    __syncthreads();
    for (int slab = wave; slab < T / 16; slab += NW) {
        if (slab != wave) {
            const float* qp = base + tok(slab * 16 + ln) + lk * DQ;
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) qf[i] = *reinterpret_cast<const float4*>(qp + 4 * i);
        }
        // ---- S^T = K Q^T: tile cb = keys cb * 16 .. + 15 (rows) x the slab's 16 queries (columns) -----------------------------
        f32x4 s[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            s[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* kp = &sK[(cb * 16 + ln) * KSTR + lk * DQ];
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + 4 * i);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[i].x, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[i].y, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[i].z, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[i].w, s[cb], 0, 0, 0);
            }
            if constexpr (NCB >= 16) __builtin_amdgcn_sched_barrier(0);   // (the scheduler hoists every tile's K reads and spills)
        }
        // ---- scale, bias, softmax over the T keys of query ln: this lane's 4 NCB values + the other three lane groups ------------
        float mx = -INFINITY;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + (size_t)(slab * 16 + ln) * T + cb * 16 + lk * 4);
            s[cb][0] = s[cb][0] * scale + bv.x; s[cb][1] = s[cb][1] * scale + bv.y;
            s[cb][2] = s[cb][2] * scale + bv.z; s[cb][3] = s[cb][3] * scale + bv.w;
            mx = fmaxf(fmaxf(mx, fmaxf(s[cb][0], s[cb][1])), fmaxf(s[cb][2], s[cb][3]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(s[cb][r] - mx);
                s[cb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        // ---- O^T = V^T P^T: reduction step (cb, r) = key cb * 16 + lk * 4 + r; B = the (unnormalised) probabilities as they lie ------
        f32x4 o[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) o[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = &sV[(cb * 16 + lk * 4 + r) * VSTR + ln];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    o[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[nb * 16], s[cb][r], o[nb], 0, 0, 0);
                if constexpr (NCB >= 16) { if (r == 3) __builtin_amdgcn_sched_barrier(0); }
            }
        // ---- D layout of O^T: channels nb * 16 + lk * 4 + {0..3} of query ln -> one 16-B store per tile ------------------------------
        {
            const int t = slab * 16 + ln;
            const int y = ih * WS + t / WS, x = iw * WS + t % WS;
            float* op = out + (((size_t)l * H + y) * W + x) * MD + (size_t)h * D + lk * 4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                *reinterpret_cast<float4*>(op + nb * 16) = make_float4(o[nb][0] * inv, o[nb][1] * inv, o[nb][2] * inv, o[nb][3] * inv);
        }
    }
}

// ---- backward (training; OPT-IN until measured: HEAL_WATTN_GRAD=kernel) -----------------------------------------------------------
// What autograd derives from mswin.py:64-78 (scores + bias -> softmax -> weighted sum), as two passes over the same blocks:
//   pass A (keys / values of the window in LDS, a wave owns 16-QUERY slabs: the forward's loop) recomputes P, then per key tile
//       dP^T = V dO^T,  dS^T = P^T o (dP^T - D),  D_q = dO_q . O_q (from the saved output: no second sweep),
//       dQ^T += K^T dS^T  (the forward's O^T pattern with K for V and dS for P), grad_bias += dS (atomics), and leaves
//       (max, 1 / sum, D) per query in `stats`;
//   pass B (QUERIES and dO of the window in LDS, a wave owns 16-KEY slabs) rebuilds one score tile at a time from the stats,
//       dV^T += dO^T P,  dK^T += Q^T dS.
// Every GEMM is one of the forward's two MFMA patterns, so no operand changes layout here either.
template <int WS, int D>
__global__ __launch_bounds__(64 * wattn_waves<WS>()) void k_window_attn_bwd_q(
    const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ outp /*[L,H,W,m*D]*/,
    const float* __restrict__ gout /*[L,H,W,m*D]*/, int H, int W, int m, float scale, float* __restrict__ gqkv,
    float* __restrict__ gbias /*[T,T] or null*/, float4* __restrict__ stats /*[L][windows][m][T]*/) {
    constexpr int T = WS * WS, NCB = T / 16, NB = D / 16, NW = wattn_waves<WS>();
    constexpr int DQ = D / 4;
    constexpr int KSTR = D + 4, VSTR = D + 4;
    __shared__ __attribute__((aligned(16))) float sK[T * KSTR];
    __shared__ __attribute__((aligned(16))) float sV[T * VSTR];
    const int nww = W / WS;
    const int ih = blockIdx.x / nww, iw = blockIdx.x - ih * nww, h = blockIdx.y, l = blockIdx.z;
    const int MD = m * D;
    const size_t C3 = (size_t)3 * MD;
    const float* base = qkv + ((size_t)l * H * W) * C3 + (size_t)h * D;
    auto pix = [&](int t) -> size_t {          // token t of this window -> pixel index inside the agent's map
        const int y = ih * WS + t / WS, x = iw * WS + t % WS;
        return (size_t)y * W + x;
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lk = lane >> 4, ln = lane & 15;
    for (int e = threadIdx.x; e < T * (D / 4); e += 64 * NW) {
        const int t = e / (D / 4), c4 = e - t * (D / 4);
        const float* row = base + pix(t) * C3;
        *reinterpret_cast<float4*>(&sK[t * KSTR + c4 * 4]) = *reinterpret_cast<const float4*>(row + MD + c4 * 4);
        *reinterpret_cast<float4*>(&sV[t * VSTR + c4 * 4]) = *reinterpret_cast<const float4*>(row + 2 * MD + c4 * 4);
    }
    __syncthreads();
    float4* st = stats + (((size_t)l * gridDim.x + blockIdx.x) * m + h) * T;
    for (int slab = wave; slab < T / 16; slab += NW) {
        const size_t qpix = (size_t)l * H * W + pix(slab * 16 + ln);
        float4 qf[DQ / 4], gf[DQ / 4];
        float dsum = 0.f;
        {
            const float* qp = qkv + qpix * C3 + (size_t)h * D + lk * DQ;
            const float* gp = gout + qpix * MD + (size_t)h * D + lk * DQ;
            const float* op = outp + qpix * MD + (size_t)h * D + lk * DQ;
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                qf[i] = *reinterpret_cast<const float4*>(qp + 4 * i);
                gf[i] = *reinterpret_cast<const float4*>(gp + 4 * i);
                const float4 of = *reinterpret_cast<const float4*>(op + 4 * i);
                dsum += (gf[i].x * of.x + gf[i].y * of.y) + (gf[i].z * of.z + gf[i].w * of.w);
            }
        }
        dsum += __shfl_xor(dsum, 16, 64);      // the four lane groups hold the four quarters of the channels of query ln
        dsum += __shfl_xor(dsum, 32, 64);
        // ---- S^T = K Q^T, scale, bias, softmax: the forward, normalised -----------------------------------------------------
        f32x4 s[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            s[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* kp = &sK[(cb * 16 + ln) * KSTR + lk * DQ];
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                const float4 kf = *reinterpret_cast<const float4*>(kp + 4 * i);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[i].x, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[i].y, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[i].z, s[cb], 0, 0, 0);
                s[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[i].w, s[cb], 0, 0, 0);
            }
            if constexpr (NCB >= 16) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4*>(bias + (size_t)(slab * 16 + ln) * T + cb * 16 + lk * 4);
            s[cb][0] = s[cb][0] * scale + bv.x; s[cb][1] = s[cb][1] * scale + bv.y;
            s[cb][2] = s[cb][2] * scale + bv.z; s[cb][3] = s[cb][3] * scale + bv.w;
            mx = fmaxf(fmaxf(mx, fmaxf(s[cb][0], s[cb][1])), fmaxf(s[cb][2], s[cb][3]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(s[cb][r] - mx);
                s[cb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (lk == 0) st[slab * 16 + ln] = make_float4(mx, inv, dsum, 0.f);
        // ---- per key tile: dP^T = V dO^T, dS^T = P^T o (dP^T - D), dQ^T += K^T dS^T ----------------------------------------------
        f32x4 dq[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dq[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x4 dp = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* vp = &sV[(cb * 16 + ln) * VSTR + lk * DQ];
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                const float4 vf = *reinterpret_cast<const float4*>(vp + 4 * i);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, gf[i].x, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, gf[i].y, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, gf[i].z, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, gf[i].w, dp, 0, 0, 0);
            }
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] = s[cb][r] * inv * (dp[r] - dsum);
            if (gbias) {
                float* gb = gbias + (size_t)(slab * 16 + ln) * T + cb * 16 + lk * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(gb + r, ds[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* kp = &sK[(cb * 16 + lk * 4 + r) * KSTR + ln];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    dq[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[nb * 16], ds[r], dq[nb], 0, 0, 0);
            }
            if constexpr (NCB >= 16) __builtin_amdgcn_sched_barrier(0);
        }
        {
            float* gp = gqkv + qpix * C3 + (size_t)h * D + lk * 4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                *reinterpret_cast<float4*>(gp + nb * 16) =
                    make_float4(dq[nb][0] * scale, dq[nb][1] * scale, dq[nb][2] * scale, dq[nb][3] * scale);
        }
    }
}

template <int WS, int D>
__global__ __launch_bounds__(64 * wattn_waves<WS>()) void k_window_attn_bwd_kv(
    const float* __restrict__ qkv, const float* __restrict__ bias, const float* __restrict__ gout, int H, int W, int m,
    float scale, const float4* __restrict__ stats, float* __restrict__ gqkv) {
    constexpr int T = WS * WS, NCB = T / 16, NB = D / 16, NW = wattn_waves<WS>();
    constexpr int DQ = D / 4;
    constexpr int QSTR = D + 4, GSTR = D + 4;
    __shared__ __attribute__((aligned(16))) float sQ[T * QSTR];
    __shared__ __attribute__((aligned(16))) float sG[T * GSTR];
    __shared__ float4 sSt[T];
    const int nww = W / WS;
    const int ih = blockIdx.x / nww, iw = blockIdx.x - ih * nww, h = blockIdx.y, l = blockIdx.z;
    const int MD = m * D;
    const size_t C3 = (size_t)3 * MD;
    auto pix = [&](int t) -> size_t {
        const int y = ih * WS + t / WS, x = iw * WS + t % WS;
        return (size_t)l * H * W + (size_t)y * W + x;
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lk = lane >> 4, ln = lane & 15;
    for (int e = threadIdx.x; e < T * (D / 4); e += 64 * NW) {
        const int t = e / (D / 4), c4 = e - t * (D / 4);
        const size_t p = pix(t);
        *reinterpret_cast<float4*>(&sQ[t * QSTR + c4 * 4]) = *reinterpret_cast<const float4*>(qkv + p * C3 + (size_t)h * D + c4 * 4);
        *reinterpret_cast<float4*>(&sG[t * GSTR + c4 * 4]) = *reinterpret_cast<const float4*>(gout + p * MD + (size_t)h * D + c4 * 4);
    }
    {
        const float4* st = stats + (((size_t)l * gridDim.x + blockIdx.x) * m + h) * T;
        for (int t = threadIdx.x; t < T; t += 64 * NW) sSt[t] = st[t];
    }
    __syncthreads();
    for (int slab = wave; slab < T / 16; slab += NW) {
        const size_t kpix = pix(slab * 16 + ln);        // this lane's KEY
        float4 kf[DQ / 4], vf[DQ / 4];
        {
            const float* kp = qkv + kpix * C3 + MD + (size_t)h * D + lk * DQ;
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                kf[i] = *reinterpret_cast<const float4*>(kp + 4 * i);
                vf[i] = *reinterpret_cast<const float4*>(kp + MD + 4 * i);
            }
        }
        f32x4 dk[NB], dv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { dk[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            // score and dP tiles of queries cb * 16 .. + 15 (rows) x the slab's 16 keys (columns): lane (lk, ln) holds queries
            // cb * 16 + lk * 4 + {0..3} of key ln
            f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* qp = &sQ[(cb * 16 + ln) * QSTR + lk * DQ];
            const float* gp = &sG[(cb * 16 + ln) * GSTR + lk * DQ];
#pragma unroll
            for (int i = 0; i < DQ / 4; ++i) {
                const float4 qv = *reinterpret_cast<const float4*>(qp + 4 * i);
                const float4 gv = *reinterpret_cast<const float4*>(gp + 4 * i);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[i].x, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[i].y, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[i].z, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[i].w, sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[i].x, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[i].y, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[i].z, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[i].w, dp, 0, 0, 0);
            }
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = cb * 16 + lk * 4 + r;
                const float4 stq = sSt[q];                                   // (max, 1 / sum, D) of query q
                const float bv = bias ? bias[(size_t)q * T + slab * 16 + ln] : 0.f;
                pr[r] = expf(sc[r] * scale + bv - stq.x) * stq.y;
                ds[r] = pr[r] * (dp[r] - stq.z);
            }
            // dV^T += dO^T P and dK^T += Q^T dS: reduction step r takes query cb * 16 + lk * 4 + r from lane group lk
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* ga = &sG[(cb * 16 + lk * 4 + r) * GSTR + ln];
                const float* qa = &sQ[(cb * 16 + lk * 4 + r) * QSTR + ln];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    dv[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[nb * 16], pr[r], dv[nb], 0, 0, 0);
                    dk[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[nb * 16], ds[r], dk[nb], 0, 0, 0);
                }
            }
            if constexpr (NCB >= 16) __builtin_amdgcn_sched_barrier(0);
        }
        {
            float* gk = gqkv + kpix * C3 + MD + (size_t)h * D + lk * 4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                *reinterpret_cast<float4*>(gk + nb * 16) =
                    make_float4(dk[nb][0] * scale, dk[nb][1] * scale, dk[nb][2] * scale, dk[nb][3] * scale);
                *reinterpret_cast<float4*>(gk + MD + nb * 16) = make_float4(dv[nb][0], dv[nb][1], dv[nb][2], dv[nb][3]);
            }
        }
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

extern "C" size_t heal_window_attention_backward_workspace(int n_agents, int H, int W, int heads) {
    return (size_t)n_agents * H * W * heads * sizeof(float4) + 256;
}

extern "C" int heal_window_attention_backward(const float* qkv, const float* pos_bias, const float* out, const float* grad_out,
                                              int n_agents, int H, int W, int heads, int dim_head, int window, float scale,
                                              float* grad_qkv, float* grad_bias, void* ws, size_t ws_bytes, void* stream) {
    HEAL_REQUIRE(n_agents >= 1 && heads >= 1 && H >= 1 && W >= 1, "window_attention_backward: bad shape");
    HEAL_REQUIRE(H % window == 0 && W % window == 0, "window_attention_backward: H, W must be multiples of the window size");
    HEAL_REQUIRE(heads <= 65535 && n_agents <= 65535, "window_attention_backward: too many heads / agents for the grid");
    HEAL_REQUIRE(qkv && out && grad_out && grad_qkv, "window_attention_backward: null pointer");
    HEAL_REQUIRE(ws && ((uintptr_t)ws & 15) == 0 && ws_bytes >= heal_window_attention_backward_workspace(n_agents, H, W, heads),
                 "window_attention_backward: workspace too small or misaligned");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((H / window) * (W / window), heads, n_agents);
    float4* stats = reinterpret_cast<float4*>(ws);
#define HEAL_WAB(WS_, D_)                                                                                                     \
    if (window == WS_ && dim_head == D_) {                                                                                    \
        constexpr int NW_ = wattn_waves<WS_>();                                                                                \
        k_window_attn_bwd_q<WS_, D_><<<grid, 64 * NW_, 0, s>>>(qkv, pos_bias, out, grad_out, H, W, heads, scale, grad_qkv,     \
                                                              grad_bias, stats);                                             \
        k_window_attn_bwd_kv<WS_, D_><<<grid, 64 * NW_, 0, s>>>(qkv, pos_bias, grad_out, H, W, heads, scale, stats, grad_qkv); \
        HEAL_LAUNCH_CHECK();                                                                                                  \
        return 0;                                                                                                             \
    }
    HEAL_WAB(4, 16) HEAL_WAB(8, 32) HEAL_WAB(16, 64) HEAL_WAB(4, 32) HEAL_WAB(8, 16) HEAL_WAB(8, 64) HEAL_WAB(4, 64)
#undef HEAL_WAB
    return set_error("window_attention_backward: window %d with dim_head %d is not instantiated", window, dim_head);
}
