// K6 -- per-pixel multi-head attention over the agents of a scene.
//
// Reference arithmetic: opencood/models/sub_modules/hmsa.py:110-151 (HGTCavAttention.forward: per
// BEV pixel an L x L attention between agents, per head, keys of padded agents masked to -inf) with
// the heterogeneous relation matrices folded into the projections by the host (they are constant
// per (type_i, type_j) and HEAL always runs with a single agent type, hmsa.py:118-125), and
// opencood/models/fuse_modules/fusion_in_one.py:14-45,126-151 (AttFusion: softmax(X X^T/sqrt(C)) X per
// pixel, one head, q = k = v).
//
// The reference materialises [B,M,H,W,L,L] score tensors and [B,M,H,W,L,L,C_head] messages through
// einsum; here one 64-lane wave owns one pixel: lane = (head, channel quad), q/k/v rows are read as
// coalesced 16 B/lane loads, the L x L scores are reduced across the lanes of a head with xor
// shuffles, softmax and the weighted sum stay in registers.  HBM traffic = read q,k,v once, write
// the output once.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr int AA_MAXL = 8;

template <int LANES_PER_HEAD, int L, bool AM /*agent-major tensors [L][n_pix][C] instead of [n_pix][L][C]*/>
__global__ __launch_bounds__(256) void k_agent_attn(const float4* __restrict__ q, const float4* __restrict__ k,
                                                   const float4* __restrict__ v,
                                                   const int* __restrict__ key_mask /*[L] or null*/,
                                                   int n_pix, float scale, int out_rows,
                                                   float4* __restrict__ out) {
    // tensors are [n_pix][L][C] (or [L][n_pix][C]: the token order of the V2X-ViT GEMMs around this kernel; a pixel's row of
    // one agent is 1 KiB contiguous either way) with C = 256 = 64 lanes x 4 channels
#define HEAL_AA_ROW(p_, a_, na_) (AM ? ((size_t)(a_) * n_pix + (p_)) : ((size_t)(p_) * (na_) + (a_)))
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= n_pix) return;
    const int l = threadIdx.x & 63;
    float4 kk[L], vv[L];
    bool valid[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        kk[j] = k[HEAL_AA_ROW(pix, j, L) * 64 + l];
        vv[j] = v[HEAL_AA_ROW(pix, j, L) * 64 + l];
        valid[j] = key_mask == nullptr || key_mask[j] != 0;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (i >= out_rows) break;
        const float4 qi = q[HEAL_AA_ROW(pix, i, L) * 64 + l];
        float s[L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float d = qi.x * kk[j].x + qi.y * kk[j].y + qi.z * kk[j].z + qi.w * kk[j].w;
#pragma unroll
            for (int o = LANES_PER_HEAD / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            s[j] = valid[j] ? d * scale : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const float p = s[j] / den;
            acc.x += p * vv[j].x; acc.y += p * vv[j].y; acc.z += p * vv[j].z; acc.w += p * vv[j].w;
        }
        out[HEAL_AA_ROW(pix, i, out_rows) * 64 + l] = acc;
    }
#undef HEAL_AA_ROW
}

template <int LPH>
static int launch_l(int L, const float* q, const float* k, const float* v, const int* mask, int n_pix, float scale,
                    int out_rows, float* out, int agent_major, hipStream_t s) {
    const int blocks = ceil_div(n_pix, 4);
    const float4 *q4 = (const float4*)q, *k4 = (const float4*)k, *v4 = (const float4*)v;
    float4* o4 = (float4*)out;
#define HEAL_AA(LL)                                                                                                  \
    case LL:                                                                                                         \
        if (agent_major) k_agent_attn<LPH, LL, true><<<blocks, 256, 0, s>>>(q4, k4, v4, mask, n_pix, scale, out_rows, o4); \
        else k_agent_attn<LPH, LL, false><<<blocks, 256, 0, s>>>(q4, k4, v4, mask, n_pix, scale, out_rows, o4);      \
        break;
    switch (L) {
        HEAL_AA(1) HEAL_AA(2) HEAL_AA(3) HEAL_AA(4) HEAL_AA(5) HEAL_AA(6) HEAL_AA(7) HEAL_AA(8)
        default: return set_error("agent_attention: L must be in [1,%d]", AA_MAXL);
    }
#undef HEAL_AA
    HEAL_LAUNCH_CHECK();
    return 0;
}

// ---- backward (training: hmsa.py:110-151 / fusion_in_one.py:14-45 under autograd) --------------------------------------------
// Same mapping as the forward: one wave per pixel, lane = (head, channel quad).  The probabilities are recomputed from q and k
// (nothing but q, k, v is saved by the forward); per query row i
//     p = softmax(s),  dp_j = dO_i . v_j,  ds_j = p_j (dp_j - sum_j p_j dp_j) * scale,
//     dq_i = sum_j ds_j k_j,   dk_j += ds_j q_i,   dv_j += p_j dO_i
// -- the two dot products are the only cross-lane steps (xor shuffles inside a head's lanes); dq / dk / dv are lane-local.
// grad_out holds rows 0 .. out_rows-1 (AttFusion keeps the ego row: the other rows' gradient is zero and their dq is written 0).
template <int LANES_PER_HEAD, int L, bool AM>
__global__ __launch_bounds__(256) void k_agent_attn_bwd(const float4* __restrict__ q, const float4* __restrict__ k,
                                                       const float4* __restrict__ v, const int* __restrict__ key_mask,
                                                       const float4* __restrict__ gout, int n_pix, float scale, int out_rows,
                                                       float4* __restrict__ gq, float4* __restrict__ gk,
                                                       float4* __restrict__ gv) {
#define HEAL_AA_ROW(p_, a_, na_) (AM ? ((size_t)(a_) * n_pix + (p_)) : ((size_t)(p_) * (na_) + (a_)))
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= n_pix) return;
    const int l = threadIdx.x & 63;
    float4 kk[L], vv[L], dk[L], dv[L];
    bool valid[L];
#pragma unroll
    for (int j = 0; j < L; ++j) {
        kk[j] = k[HEAL_AA_ROW(pix, j, L) * 64 + l];
        vv[j] = v[HEAL_AA_ROW(pix, j, L) * 64 + l];
        dk[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        dv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        valid[j] = key_mask == nullptr || key_mask[j] != 0;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (i >= out_rows) {
            gq[HEAL_AA_ROW(pix, i, L) * 64 + l] = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const float4 qi = q[HEAL_AA_ROW(pix, i, L) * 64 + l];
        const float4 go = gout[HEAL_AA_ROW(pix, i, out_rows) * 64 + l];
        float s[L], dp[L];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < L; ++j) {
            float d = qi.x * kk[j].x + qi.y * kk[j].y + qi.z * kk[j].z + qi.w * kk[j].w;
            float e = go.x * vv[j].x + go.y * vv[j].y + go.z * vv[j].z + go.w * vv[j].w;
#pragma unroll
            for (int o = LANES_PER_HEAD / 2; o > 0; o >>= 1) { d += __shfl_xor(d, o, 64); e += __shfl_xor(e, o, 64); }
            s[j] = valid[j] ? d * scale : -INFINITY;
            dp[j] = e;
            mx = fmaxf(mx, s[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) { s[j] = s[j] / den; dsum += s[j] * dp[j]; }
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const float pj = s[j], ds = pj * (dp[j] - dsum) * scale;
            dq.x += ds * kk[j].x; dq.y += ds * kk[j].y; dq.z += ds * kk[j].z; dq.w += ds * kk[j].w;
            dk[j].x += ds * qi.x; dk[j].y += ds * qi.y; dk[j].z += ds * qi.z; dk[j].w += ds * qi.w;
            dv[j].x += pj * go.x; dv[j].y += pj * go.y; dv[j].z += pj * go.z; dv[j].w += pj * go.w;
        }
        gq[HEAL_AA_ROW(pix, i, L) * 64 + l] = dq;
    }
#pragma unroll
    for (int j = 0; j < L; ++j) {
        gk[HEAL_AA_ROW(pix, j, L) * 64 + l] = dk[j];
        gv[HEAL_AA_ROW(pix, j, L) * 64 + l] = dv[j];
    }
#undef HEAL_AA_ROW
}

template <int LPH>
static int launch_bwd_l(int L, const float* q, const float* k, const float* v, const int* mask, const float* gout, int n_pix,
                        float scale, int out_rows, float* gq, float* gk, float* gv, int agent_major, hipStream_t s) {
    const int blocks = ceil_div(n_pix, 4);
    const float4 *q4 = (const float4*)q, *k4 = (const float4*)k, *v4 = (const float4*)v, *g4 = (const float4*)gout;
    float4 *gq4 = (float4*)gq, *gk4 = (float4*)gk, *gv4 = (float4*)gv;
#define HEAL_AAB(LL)                                                                                                                \
    case LL:                                                                                                                        \
        if (agent_major) k_agent_attn_bwd<LPH, LL, true><<<blocks, 256, 0, s>>>(q4, k4, v4, mask, g4, n_pix, scale, out_rows, gq4, gk4, gv4); \
        else k_agent_attn_bwd<LPH, LL, false><<<blocks, 256, 0, s>>>(q4, k4, v4, mask, g4, n_pix, scale, out_rows, gq4, gk4, gv4);  \
        break;
    switch (L) {
        HEAL_AAB(1) HEAL_AAB(2) HEAL_AAB(3) HEAL_AAB(4) HEAL_AAB(5) HEAL_AAB(6) HEAL_AAB(7) HEAL_AAB(8)
        default: return set_error("agent_attention_backward: L must be in [1,%d]", AA_MAXL);
    }
#undef HEAL_AAB
    HEAL_LAUNCH_CHECK();
    return 0;
}

}  // namespace heal

using namespace heal;

extern "C" int heal_agent_attention(const float* q, const float* k, const float* v, const int32_t* key_mask,
                                    int n_pix, int n_agents, int channels, int heads, float scale, int out_rows,
                                    float* out, int agent_major, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(channels == 256, "agent_attention: channels must be 256 (got %d)", channels);
    HEAL_REQUIRE(out_rows >= 1 && out_rows <= n_agents, "agent_attention: out_rows must be in [1,n_agents]");
    if (n_pix <= 0) return 0;
    switch (heads) {
        case 1: return launch_l<64>(n_agents, q, k, v, key_mask, n_pix, scale, out_rows, out, agent_major, s);
        case 4: return launch_l<16>(n_agents, q, k, v, key_mask, n_pix, scale, out_rows, out, agent_major, s);
        case 8: return launch_l<8>(n_agents, q, k, v, key_mask, n_pix, scale, out_rows, out, agent_major, s);
        case 16: return launch_l<4>(n_agents, q, k, v, key_mask, n_pix, scale, out_rows, out, agent_major, s);
        default: return set_error("agent_attention: heads must be 1, 4, 8 or 16 (got %d)", heads);
    }
}

extern "C" int heal_agent_attention_backward(const float* q, const float* k, const float* v, const int32_t* key_mask,
                                             const float* grad_out, int n_pix, int n_agents, int channels, int heads, float scale,
                                             int out_rows, float* grad_q, float* grad_k, float* grad_v, int agent_major,
                                             void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(channels == 256, "agent_attention_backward: channels must be 256 (got %d)", channels);
    HEAL_REQUIRE(out_rows >= 1 && out_rows <= n_agents, "agent_attention_backward: out_rows must be in [1,n_agents]");
    if (n_pix <= 0) return 0;
    switch (heads) {
        case 1: return launch_bwd_l<64>(n_agents, q, k, v, key_mask, grad_out, n_pix, scale, out_rows, grad_q, grad_k, grad_v, agent_major, s);
        case 4: return launch_bwd_l<16>(n_agents, q, k, v, key_mask, grad_out, n_pix, scale, out_rows, grad_q, grad_k, grad_v, agent_major, s);
        case 8: return launch_bwd_l<8>(n_agents, q, k, v, key_mask, grad_out, n_pix, scale, out_rows, grad_q, grad_k, grad_v, agent_major, s);
        case 16: return launch_bwd_l<4>(n_agents, q, k, v, key_mask, grad_out, n_pix, scale, out_rows, grad_q, grad_k, grad_v, agent_major, s);
        default: return set_error("agent_attention_backward: heads must be 1, 4, 8 or 16 (got %d)", heads);
    }
}
