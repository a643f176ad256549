// K5 -- warp every agent's BEV features and occupancy score into the ego frame and fuse them with a
// per-pixel softmax over agents.
//
// Reference arithmetic (one pyramid level, one scene):
//   opencood/models/fuse_modules/pyramid_fuse.py:17-63   weighted_fuse
//   opencood/models/fuse_modules/pyramid_fuse.py:145-162 score = sigmoid(occ) + 1e-4, eval-mode
//                                                         camera crop mask
//   opencood/models/sub_modules/torch_transformation_utils.py:323-332  warp_affine_simple =
//       F.affine_grid(M, size, align_corners=False) (+ `.to(src)`) and
//       F.grid_sample(bilinear, zeros padding, align_corners=False)
// The reference runs ~8 ATen kernels per level, each streaming the whole agent stack
// (warp features, warp scores, compare, masked_fill, softmax, isnan/where, multiply, sum) and
// materialises the warped stack.  Here one kernel reads each source tap once and writes the fused
// map once: bytes moved = the operator's compulsory traffic.
//
// Sampling arithmetic follows PyTorch operation for operation: base grid = linspace(-1,1,W)*(W-1)/W
// in the dtype of the affine matrix (float64 when pairwise_t_matrix comes from numpy), grid =
// base @ M^T, rounded to fp32, unnormalise ((g+1)*size-1)/2, floor, corner weights as products of
// fp32 differences, taps outside the image contribute zero.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr int WF_MAXA = 8;     // agents handled per launch (max_cav is 5..8 in the reference configs)
constexpr int WF_TW = 32, WF_TH = 8;  // pixel tile per 256-thread block

struct WarpParams {
    const double* mdev;     // device copy of the affine rows [n_agents][6] (wins over `m` when non-null: a captured HIP
                            // graph then follows the poses of the frame it is replayed on instead of the captured ones)
    double m[WF_MAXA][6];   // rows of affine_matrix[b][0, a]: m00 m01 m02 m10 m11 m12
    int crop[WF_MAXA][4];   // (h0,h1,w0,w1) window where the score is kept; h1<=h0: keep everything
    int n_agents, C, H, W;
    int grid_f64;
};

struct Taps {
    int off;       // y0*W + x0 (may point outside; guarded by `ok`)
    float w[4];    // nw, ne, sw, se
    unsigned ok;   // bit k set: tap k lies inside the image
};

template <typename T>
__device__ __forceinline__ T base_coord(int j, int n) {
    // torch.linspace(-1, 1, n) * (n - 1) / n, element j
    if (n <= 1) return (T)0;
    const T step = (T)2 / (T)(n - 1);
    const T v = (j < n / 2) ? (T)-1 + step * (T)j : (T)1 - step * (T)(n - 1 - j);
    return v * (T)(n - 1) / (T)n;
}

__device__ __forceinline__ void load_affine(const WarpParams& p, int a, double (&m)[6]) {
    // wave-uniform: six scalar loads from the kernel arguments or from the device buffer
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = p.mdev ? p.mdev[a * 6 + k] : p.m[a][k];
}

template <typename T>
__device__ __forceinline__ void grid_point(const double* m, int h, int w, int H, int W, float& gx, float& gy) {
    const T xs = base_coord<T>(w, W), ys = base_coord<T>(h, H);
    gx = (float)(((T)m[0] * xs + (T)m[1] * ys) + (T)m[2]);
    gy = (float)(((T)m[3] * xs + (T)m[4] * ys) + (T)m[5]);
}

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
    const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    const float x0 = floorf(ix), y0 = floorf(iy);
    const float x1 = x0 + 1.f, y1 = y0 + 1.f;
    Taps t;
    t.w[0] = (x1 - ix) * (y1 - iy);
    t.w[1] = (ix - x0) * (y1 - iy);
    t.w[2] = (x1 - ix) * (iy - y0);
    t.w[3] = (ix - x0) * (iy - y0);
    const bool xin0 = x0 >= 0.f && x0 <= (float)(W - 1);
    const bool xin1 = x1 >= 0.f && x1 <= (float)(W - 1);
    const bool yin0 = y0 >= 0.f && y0 <= (float)(H - 1);
    const bool yin1 = y1 >= 0.f && y1 <= (float)(H - 1);
    t.ok = (unsigned)(xin0 && yin0) | ((unsigned)(xin1 && yin0) << 1) | ((unsigned)(xin0 && yin1) << 2) |
           ((unsigned)(xin1 && yin1) << 3);
    // offsets are only dereferenced for taps with their bit set; keep the int conversion defined
    const float xc = fminf(fmaxf(x0, -2.f), (float)W), yc = fminf(fmaxf(y0, -2.f), (float)H);
    t.off = (int)yc * W + (int)xc;
    return t;
}

__device__ __forceinline__ float score_at(const float* __restrict__ occ, int idx, int W, const int* crop) {
    // sigmoid(occ) + 1e-4, zero outside the camera crop window (pyramid_fuse.py:145-162)
    if (crop[1] > crop[0]) {
        const int h = idx / W, w = idx - h * W;
        if (h < crop[0] || h >= crop[1] || w < crop[2] || w >= crop[3]) return 0.f;
    }
    return 1.f / (1.f + expf(-occ[idx])) + 1e-4f;
}

__device__ __forceinline__ float sample(const float* __restrict__ src, const Taps& t, int W) {
    // nw, ne, sw, se accumulated in that order (taps outside contribute exactly zero)
    float acc = 0.f;
    if (t.ok & 1u) acc = src[t.off] * t.w[0];
    if (t.ok & 2u) acc += src[t.off + 1] * t.w[1];
    if (t.ok & 4u) acc += src[t.off + W] * t.w[2];
    if (t.ok & 8u) acc += src[t.off + W + 1] * t.w[3];
    return acc;
}

__device__ __forceinline__ float sample_score(const float* __restrict__ occ, const Taps& t, int W, const int* crop) {
    float acc = 0.f;
    if (t.ok & 1u) acc = score_at(occ, t.off, W, crop) * t.w[0];
    if (t.ok & 2u) acc += score_at(occ, t.off + 1, W, crop) * t.w[1];
    if (t.ok & 4u) acc += score_at(occ, t.off + W, W, crop) * t.w[2];
    if (t.ok & 8u) acc += score_at(occ, t.off + W + 1, W, crop) * t.w[3];
    return acc;
}

// softmax over agents of the warped scores with the reference's masking rules:
// score == 0 -> -inf; all agents masked -> NaN -> 0 (pyramid_fuse.py:51-58)
__device__ __forceinline__ void agent_softmax(float* s, int n) {
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < WF_MAXA; ++a)
        if (a < n) { if (s[a] == 0.f) s[a] = -INFINITY; mx = fmaxf(mx, s[a]); }
    if (mx == -INFINITY) {
#pragma unroll
        for (int a = 0; a < WF_MAXA; ++a) s[a] = 0.f;
        return;
    }
    float den = 0.f;
#pragma unroll
    for (int a = 0; a < WF_MAXA; ++a)
        if (a < n) { s[a] = expf(s[a] - mx); den += s[a]; }
#pragma unroll
    for (int a = 0; a < WF_MAXA; ++a)
        if (a < n) s[a] = s[a] / den;
}

// ---- fused: warp all agents + softmax + weighted sum -------------------------------------------
// Per agent the four taps become (offset, weight) pairs with the softmax probability and the
// "inside the image" test folded into the weight (outside taps: weight 0, offset clamped to 0), so the
// channel loop is branch-free: 4 channels x n_agents x 4 taps independent loads are in flight per thread.
template <int NA>
__global__ __launch_bounds__(256, 4) void k_warp_fuse(const float* __restrict__ feats,
                                                     const float* __restrict__ occ, WarpParams p, int CCH,
                                                     float* __restrict__ out) {
    const Block3 bk = xcd_block();  // rotated bilinear footprints of neighbouring tiles overlap: keep them on one L2
    const int w = bk.x * WF_TW + (threadIdx.x & (WF_TW - 1));
    const int h = bk.y * WF_TH + (threadIdx.x / WF_TW);
    if (w >= p.W || h >= p.H) return;
    const int HW = p.H * p.W;
    int off[NA][4];
    float wt[NA][4];
    float prob[WF_MAXA];
#pragma unroll
    for (int a = NA; a < WF_MAXA; ++a) prob[a] = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        prob[a] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { off[a][k] = 0; wt[a][k] = 0.f; }
        {
            float gx, gy;
            double m[6];
            load_affine(p, a, m);
            if (p.grid_f64) grid_point<double>(m, h, w, p.H, p.W, gx, gy);
            else grid_point<float>(m, h, w, p.H, p.W, gx, gy);
            const Taps t = make_taps(gx, gy, p.H, p.W);
            prob[a] = sample_score(occ + (size_t)a * HW, t, p.W, p.crop[a]);
            const int o4[4] = {t.off, t.off + 1, t.off + p.W, t.off + p.W + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = (t.ok >> k) & 1u;
                off[a][k] = ok ? o4[k] : 0;
                wt[a][k] = ok ? t.w[k] : 0.f;
            }
        }
    }
    agent_softmax(prob, NA);
    unsigned live = 0;       // bit a: some lane of this wave samples agent a inside its map (else every tap weight is zero)
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wt[a][k] *= prob[a];
        const bool any = (wt[a][0] != 0.f) | (wt[a][1] != 0.f) | (wt[a][2] != 0.f) | (wt[a][3] != 0.f);
        live |= __ballot(any) ? (1u << a) : 0u;
    }
    const int c0 = bk.z * CCH;
    const int pix = h * p.W + w;
    constexpr int U = NA <= 4 ? 4 : (NA <= 6 ? 2 : 1);  // channels in flight per thread (register budget: 4 waves per SIMD)
    for (int c = c0; c < c0 + CCH && c < p.C; c += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if ((live >> a) & 1u) {   // wave-uniform: an agent whose map does not reach this tile costs no loads
                const float* base = feats + ((size_t)a * p.C + c) * HW;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c + u < p.C) {
                        const float* src = base + (size_t)u * HW;
                        float v = src[off[a][0]] * wt[a][0];
                        v += src[off[a][1]] * wt[a][1];
                        v += src[off[a][2]] * wt[a][2];
                        v += src[off[a][3]] * wt[a][3];
                        acc[u] += v;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c + u < p.C) out[(size_t)(c + u) * HW + pix] = acc[u];
    }
}

// ---- round 4: all pyramid levels in ONE launch, source footprints staged through LDS ------------------------------------------
// k_warp_fuse above gathers every tap straight from global memory: one thread = one ego pixel, and for each channel the 64
// lanes of a wave read 4 x n_agents scattered 4-byte words along a rotated line of the source map -- 16-32 cache lines per wave
// instruction.  Its HBM traffic equals the algorithmic bytes (PMC), but the texture-address path is the limit: 35 cycles per
// gather instruction per CU, 0.16 of the HBM roof.  Here a block is a 16 x 16 tile of ego pixels x a slice of the channels:
//   1. every thread computes, once, its taps per agent (the reference's sampling arithmetic: make_taps), samples the occupancy
//      score (a one-channel map: 4 words per agent), runs the masked softmax over agents and folds the probability and the
//      "inside the image" test into four tap weights per agent;
//   2. per agent the block reduces the bounding box of the taps it will read (LDS min / max): for a rigid pose <= 25 x 25 source
//      pixels for 256 ego pixels;
//   3. per agent and group of 8 channels that box goes global -> LDS as ROW SEGMENTS (a half-wave reads up to 32 consecutive
//      floats: 2-4 cache lines per wave instruction instead of 16-32), and every thread takes its four taps from LDS;
//   4. a box that does not fit the 32 x 32 staging tile (a zooming affine matrix) falls back to the direct gather for that
//      agent -- any matrix gives the right answer.
// Measured and dropped: a software pipeline over the (round, agent) slots with two LDS buffers and one barrier per slot (the
// next slot's row segments in flight under the current slot's taps): 115 vs 85 us -- two resident blocks per CU instead of four
// lose more than the halved barrier count gains.
// All levels of the pyramid are one grid (level = a block-index range): one launch instead of three, no launch gaps on the
// critical path of the scene (pyramid_fuse.py:104-168 runs the three levels back to back).
constexpr int WL_T = 16;            // ego tile side
constexpr int WL_BMAX = 32;         // largest staged source box side
constexpr int WL_CC = 8;            // channels per staging round
constexpr int WL_MAXL = 4;          // pyramid levels per launch

struct WfLevel {
    const float* feats;     // [n_agents, C, H, W]
    const float* occ;       // [n_agents, 1, H, W]
    float* out;             // [C, H, W]
    int C, H, W;
    int tiles_x, cgroups, cpb;   // tiles per row, channel slices per tile, channels per slice
    int block0;                  // first block of this level
    int crop[WF_MAXA][4];
};
struct WfLevels {
    WfLevel lv[WL_MAXL];
    int n_levels, n_agents, grid_f64, dbg;
    const double* mdev;
    double m[WF_MAXA][6];
};

template <int NA>
__global__ __launch_bounds__(256) void k_warp_fuse_lds(const WfLevels P) {
    __shared__ float s_box_f[WL_CC * WL_BMAX * (WL_BMAX + 1)];
    __shared__ int s_box[WF_MAXA][4];      // min x, min y, max x, max y of the taps the block reads from agent a
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < WL_MAXL; ++i)
        if (i < P.n_levels && (int)blockIdx.x >= P.lv[i].block0) lvl = i;
    const WfLevel& L = P.lv[lvl];
    const int lb = blockIdx.x - L.block0;
    const int cg = lb % L.cgroups, tile = lb / L.cgroups;
    const int tx0 = (tile % L.tiles_x) * WL_T, ty0 = (tile / L.tiles_x) * WL_T;
    const int tid = threadIdx.x;
    const int w = tx0 + (tid & (WL_T - 1)), h = ty0 + (tid >> 4);
    const bool live = w < L.W && h < L.H;
    const int HW = L.H * L.W;
    if (tid < NA) { s_box[tid][0] = 1 << 30; s_box[tid][1] = 1 << 30; s_box[tid][2] = -(1 << 30); s_box[tid][3] = -(1 << 30); }
    // base grid coordinates of the tile's 16 columns and 16 rows (torch.linspace(-1, 1, n) * (n - 1) / n: two divisions each, in
    // fp64 when the affine matrix is): evaluated once by 32 threads instead of twice by all 256
    __shared__ double s_base_d[2][WL_T];
    __shared__ float s_base_f[2][WL_T];
    if (tid < 2 * WL_T) {
        const int ax = tid >> 4, j = tid & (WL_T - 1);
        const int idx = (ax ? ty0 : tx0) + j, nn = ax ? L.H : L.W;
        s_base_d[ax][j] = base_coord<double>(min(idx, nn - 1), nn);
        s_base_f[ax][j] = base_coord<float>(min(idx, nn - 1), nn);
    }
    __syncthreads();

    int tx[NA], ty[NA];          // north-west tap of agent a (clamped into [-2, W] x [-2, H]: only dereferenced under `ok`)
    unsigned okb[NA];
    float wt[NA][4];
    float prob[WF_MAXA];
#pragma unroll
    for (int a = NA; a < WF_MAXA; ++a) prob[a] = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        prob[a] = 0.f; okb[a] = 0u; tx[a] = 0; ty[a] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) wt[a][k] = 0.f;
        if (live) {
            float gx, gy;
            double m[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) m[k] = P.mdev ? P.mdev[a * 6 + k] : P.m[a][k];   // wave-uniform scalar loads
            if (P.grid_f64) {      // grid_point<T> with the base coordinates taken from LDS (same operations, same order)
                const double xs = s_base_d[0][tid & (WL_T - 1)], ys = s_base_d[1][tid >> 4];
                gx = (float)((m[0] * xs + m[1] * ys) + m[2]);
                gy = (float)((m[3] * xs + m[4] * ys) + m[5]);
            } else {
                const float xs = s_base_f[0][tid & (WL_T - 1)], ys = s_base_f[1][tid >> 4];
                gx = (float)(((float)m[0] * xs + (float)m[1] * ys) + (float)m[2]);
                gy = (float)(((float)m[3] * xs + (float)m[4] * ys) + (float)m[5]);
            }
            const Taps t = make_taps(gx, gy, L.H, L.W);
            prob[a] = sample_score(L.occ + (size_t)a * HW, t, L.W, L.crop[a]);
            // north-west tap coordinates: the clamps of make_taps on the coordinates themselves (t.off = y0 * W + x0)
            const float ix = ((gx + 1.f) * (float)L.W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)L.H - 1.f) / 2.f;
            tx[a] = (int)fminf(fmaxf(floorf(ix), -2.f), (float)L.W);
            ty[a] = (int)fminf(fmaxf(floorf(iy), -2.f), (float)L.H);
            okb[a] = t.ok;
#pragma unroll
            for (int k = 0; k < 4; ++k) wt[a][k] = ((t.ok >> k) & 1u) ? t.w[k] : 0.f;
        }
    }
    agent_softmax(prob, NA);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wt[a][k] *= prob[a];
        // taps that carry weight extend the agent's box
        const bool any = (wt[a][0] != 0.f) | (wt[a][1] != 0.f) | (wt[a][2] != 0.f) | (wt[a][3] != 0.f);
        if (!any) okb[a] = 0u;
        // wave-level min / max first, then ONE LDS atomic per wave and bound: 256 threads hitting the same four words serialise
        // (measured: the prologue alone took 200 us of the kernel's 250 with per-thread atomics)
        int x_lo = any ? ((okb[a] & 5u) ? tx[a] : tx[a] + 1) : (1 << 30), x_hi = any ? ((okb[a] & 10u) ? tx[a] + 1 : tx[a]) : -(1 << 30);
        int y_lo = any ? ((okb[a] & 3u) ? ty[a] : ty[a] + 1) : (1 << 30), y_hi = any ? ((okb[a] & 12u) ? ty[a] + 1 : ty[a]) : -(1 << 30);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            x_lo = min(x_lo, __shfl_xor(x_lo, o, 64)); y_lo = min(y_lo, __shfl_xor(y_lo, o, 64));
            x_hi = max(x_hi, __shfl_xor(x_hi, o, 64)); y_hi = max(y_hi, __shfl_xor(y_hi, o, 64));
        }
        if ((tid & 63) == 0 && x_lo <= x_hi) {
            atomicMin(&s_box[a][0], x_lo); atomicMin(&s_box[a][1], y_lo);
            atomicMax(&s_box[a][2], x_hi); atomicMax(&s_box[a][3], y_hi);
        }
    }
    __syncthreads();

    const int c_lo = cg * L.cpb, c_hi = (P.dbg & 1) ? c_lo : min(c_lo + L.cpb, L.C);
    const int pix = h * L.W + w;
    const int lx = tid & 31, ry = tid >> 5;      // staging role: column of a box row, one of 8 row workers
    for (int c0 = c_lo; c0 < c_hi; c0 += WL_CC) {
        float acc[WL_CC];
#pragma unroll
        for (int u = 0; u < WL_CC; ++u) acc[u] = 0.f;
        const int nc = min(WL_CC, c_hi - c0);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int bx0 = s_box[a][0], by0 = s_box[a][1], bw = s_box[a][2] - bx0 + 1, bh = s_box[a][3] - by0 + 1;
            if (bw <= 0 || bh <= 0) continue;                       // block-uniform: no pixel of the tile sees agent a
            const float* __restrict__ base = L.feats + ((size_t)a * L.C + c0) * HW;
            if (bw <= WL_BMAX && bh <= WL_BMAX) {
                const int pitch = bw | 1;
                // the [nc][bh][bw] box: lane lx = column, row worker ry takes rows ry, ry + 8, ry + 16, ry + 24 of every channel.
                // ALL loads of a thread are issued (on clamped, always valid addresses) before the first LDS store: a load ->
                // store loop with a run-time trip count waits for every load in turn (the first version of this kernel: 2.3x
                // slower than the direct gather)
                float st[4][WL_CC];
                const int xs = min(lx, bw - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* rowp = base + (size_t)(by0 + min(ry + 8 * j, bh - 1)) * L.W + bx0 + xs;
#pragma unroll
                    for (int u = 0; u < WL_CC; ++u) st[j][u] = (P.dbg & 2) ? 1.f : rowp[(size_t)min(u, nc - 1) * HW];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = ry + 8 * j;
                    if (lx < bw && y < bh) {
#pragma unroll
                        for (int u = 0; u < WL_CC; ++u)
                            if (u < nc) s_box_f[(u * bh + y) * pitch + lx] = st[j][u];
                    }
                }
                __syncthreads();
                if (okb[a] && !(P.dbg & 4)) {
                    const int o = (ty[a] - by0) * pitch + (tx[a] - bx0);
                    // taps without weight may lie outside the box: clamp their offsets to a staged word (weight 0 kills them)
                    const int o0 = (okb[a] & 1u) ? o : 0, o1 = (okb[a] & 2u) ? o + 1 : 0;
                    const int o2 = (okb[a] & 4u) ? o + pitch : 0, o3 = (okb[a] & 8u) ? o + pitch + 1 : 0;
#pragma unroll
                    for (int u = 0; u < WL_CC; ++u) {
                        if (u < nc) {
                            const float* s = s_box_f + u * bh * pitch;
                            float v = s[o0] * wt[a][0];
                            v += s[o1] * wt[a][1];
                            v += s[o2] * wt[a][2];
                            v += s[o3] * wt[a][3];
                            acc[u] += v;
                        }
                    }
                }
                __syncthreads();
            } else if (okb[a]) {                                      // a box larger than the staging tile: direct gather
                const int o = ty[a] * L.W + tx[a];
                const int o0 = (okb[a] & 1u) ? o : 0, o1 = (okb[a] & 2u) ? o + 1 : 0;
                const int o2 = (okb[a] & 4u) ? o + L.W : 0, o3 = (okb[a] & 8u) ? o + L.W + 1 : 0;
#pragma unroll
                for (int u = 0; u < WL_CC; ++u) {
                    if (u < nc) {
                        const float* src = base + (size_t)u * HW;
                        float v = src[o0] * wt[a][0];
                        v += src[o1] * wt[a][1];
                        v += src[o2] * wt[a][2];
                        v += src[o3] * wt[a][3];
                        acc[u] += v;
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int u = 0; u < WL_CC; ++u)
                if (u < nc) L.out[(size_t)(c0 + u) * HW + pix] = acc[u];
        }
    }
}

// ---- split form for agent-sharded execution ------------------------------------------------------
template <int CCH>
__global__ __launch_bounds__(256) void k_warp_agent(const float* __restrict__ feat,
                                                   const float* __restrict__ occ, WarpParams p,
                                                   float* __restrict__ feat_ego,
                                                   float* __restrict__ score_ego) {
    const Block3 bk = xcd_block();
    const int w = bk.x * WF_TW + (threadIdx.x & (WF_TW - 1));
    const int h = bk.y * WF_TH + (threadIdx.x / WF_TW);
    if (w >= p.W || h >= p.H) return;
    const int HW = p.H * p.W;
    float gx, gy;
    double m[6];
    load_affine(p, 0, m);
    if (p.grid_f64) grid_point<double>(m, h, w, p.H, p.W, gx, gy);
    else grid_point<float>(m, h, w, p.H, p.W, gx, gy);
    const Taps t = make_taps(gx, gy, p.H, p.W);
    const int pix = h * p.W + w;
    if (bk.z == 0 && score_ego != nullptr) score_ego[pix] = sample_score(occ, t, p.W, p.crop[0]);
    const int c0 = bk.z * CCH;
#pragma unroll 4
    for (int c = c0; c < c0 + CCH && c < p.C; ++c)
        feat_ego[(size_t)c * HW + pix] = sample(feat + (size_t)c * HW, t, p.W);
}

// ---- warp of every agent of a scene into TOKEN-MAJOR ego-frame maps (V2X-ViT's layout) ------------------------------
// feats [n, C, H, W] -> out [n, H, W, C]: what `warp_affine_simple` + `x.permute(0, 2, 3, 1)` produce
// (fusion_in_one.py:352-358) in one pass: a block samples a 16 x 4 pixel tile of 64 channels (lanes along x: the four taps of
// neighbouring pixels share cache lines), transposes it through LDS and writes 256-byte channel runs per pixel.
constexpr int WPM_TW = 16, WPM_TH = 4, WPM_C = 64;
__global__ __launch_bounds__(256) void k_warp_agents_pm(const float* __restrict__ feats, WarpParams p,
                                                       float* __restrict__ out) {
    __shared__ float s_t[WPM_TW * WPM_TH][WPM_C + 1];
    const Block3 bk = xcd_block();
    const int cblocks = (p.C + WPM_C - 1) / WPM_C;
    const int a = bk.z / cblocks, c0 = (bk.z - a * cblocks) * WPM_C;
    const int pix_l = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int w = bk.x * WPM_TW + (pix_l & (WPM_TW - 1));
    const int h = bk.y * WPM_TH + (pix_l / WPM_TW);
    const int HW = p.H * p.W;
    if (w < p.W && h < p.H) {
        float gx, gy;
        double m[6];
        load_affine(p, a, m);
        if (p.grid_f64) grid_point<double>(m, h, w, p.H, p.W, gx, gy);
        else grid_point<float>(m, h, w, p.H, p.W, gx, gy);
        const Taps t = make_taps(gx, gy, p.H, p.W);
        const float* src = feats + ((size_t)a * p.C + c0 + cg * 16) * HW;
#pragma unroll 4
        for (int c = 0; c < 16; ++c)
            s_t[pix_l][cg * 16 + c] = (c0 + cg * 16 + c < p.C) ? sample(src + (size_t)c * HW, t, p.W) : 0.f;
    }
    __syncthreads();
    const int cq = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pl = (threadIdx.x >> 4) + 16 * i;
        const int ww = bk.x * WPM_TW + (pl & (WPM_TW - 1)), hh = bk.y * WPM_TH + (pl / WPM_TW);
        if (ww >= p.W || hh >= p.H || c0 + cq * 4 >= p.C) continue;
        const float4 v = make_float4(s_t[pl][cq * 4], s_t[pl][cq * 4 + 1], s_t[pl][cq * 4 + 2], s_t[pl][cq * 4 + 3]);
        *reinterpret_cast<float4*>(out + (((size_t)a * p.H + hh) * p.W + ww) * p.C + c0 + cq * 4) = v;
    }
}

// Agent a's maps start at feats + off.f[a] / scores + off.s[a] (float4 units): a contiguous [n, C, HW] stack, or the rows of the
// exchange buffer of the agent-sharded runner in place (no re-pack after the gather).
struct FuseOffsets { long long f[WF_MAXA], s[WF_MAXA]; };

__global__ __launch_bounds__(256) void k_fuse_warped(const float4* __restrict__ feats,
                                                    const float4* __restrict__ scores, FuseOffsets off, int n, int C,
                                                    int HW4, float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW4) return;
    float px[WF_MAXA], py[WF_MAXA], pz[WF_MAXA], pw[WF_MAXA];
#pragma unroll
    for (int a = 0; a < WF_MAXA; ++a) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a < n) s = scores[off.s[a] + i];
        px[a] = s.x; py[a] = s.y; pz[a] = s.z; pw[a] = s.w;
    }
    agent_softmax(px, n); agent_softmax(py, n); agent_softmax(pz, n); agent_softmax(pw, n);
    const int c0 = blockIdx.y * 16;
    for (int c = c0; c < c0 + 16 && c < C; ++c) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < WF_MAXA; ++a) {
            if (a < n) {
                const float4 v = feats[off.f[a] + (size_t)c * HW4 + i];
                acc.x += v.x * px[a]; acc.y += v.y * py[a]; acc.z += v.z * pz[a]; acc.w += v.w * pw[a];
            }
        }
        out[(size_t)c * HW4 + i] = acc;
    }
}

// ---- backward of the fused warp + softmax-weighted sum (training) ------------------------------------------
// out[c,p] = sum_a w_a(p) X_a[c](p),  X_a[c](p) = sum_k t_ak(p) x[a,c,off_ak],  w = masked softmax_a(S_a),
// S_a(p) = sum_k t_ak(p) s_a[off_ak],  s = (sigmoid(occ) + 1e-4) * crop.  One thread per ego pixel, all channels:
//   dL/dx[a,c,off_ak] += G[c,p] w_a t_ak                       (fp32 atomics: several ego pixels share a source pixel)
//   dL/dS_a = w_a (q_a - sum_b w_b q_b),  q_a = sum_c G[c,p] X_a[c](p)     (masked agents have w = 0: no gradient)
//   dL/docc[a,off_ak] += dL/dS_a t_ak sg (1 - sg) inside the crop window,  sg = sigmoid(occ).
// grad_feats / grad_occ must be zero on entry.  Same sampling arithmetic as the forward (make_taps, grid dtype).
template <int NA>
__global__ __launch_bounds__(256) void k_warp_fuse_backward(const float* __restrict__ feats, const float* __restrict__ occ,
                                                           WarpParams p, const float* __restrict__ gout,
                                                           float* __restrict__ gfeats, float* __restrict__ gocc) {
    const int w = blockIdx.x * WF_TW + (threadIdx.x & (WF_TW - 1));
    const int h = blockIdx.y * WF_TH + (threadIdx.x / WF_TW);
    if (w >= p.W || h >= p.H) return;
    const int HW = p.H * p.W, pix = h * p.W + w;
    int off[NA][4];
    float tw[NA][4];
    float prob[WF_MAXA], q[NA];
#pragma unroll
    for (int a = NA; a < WF_MAXA; ++a) prob[a] = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        float gx, gy;
        double m[6];
        load_affine(p, a, m);
        if (p.grid_f64) grid_point<double>(m, h, w, p.H, p.W, gx, gy);
        else grid_point<float>(m, h, w, p.H, p.W, gx, gy);
        const Taps t = make_taps(gx, gy, p.H, p.W);
        prob[a] = sample_score(occ + (size_t)a * HW, t, p.W, p.crop[a]);
        const int o4[4] = {t.off, t.off + 1, t.off + p.W, t.off + p.W + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = (t.ok >> k) & 1u;
            off[a][k] = ok ? o4[k] : -1;
            tw[a][k] = ok ? t.w[k] : 0.f;
        }
        q[a] = 0.f;
    }
    agent_softmax(prob, NA);
    for (int c = 0; c < p.C; ++c) {
        const float g = gout[(size_t)c * HW + pix];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const size_t base = ((size_t)a * p.C + c) * HW;
            float x = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (off[a][k] >= 0) {
                    x += feats[base + off[a][k]] * tw[a][k];
                    const float d = g * prob[a] * tw[a][k];
                    if (d != 0.f) unsafeAtomicAdd(gfeats + base + off[a][k], d);
                }
            q[a] = fmaf(g, x, q[a]);
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) dot = fmaf(prob[a], q[a], dot);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const float dS = prob[a] * (q[a] - dot);
        if (dS == 0.f) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (off[a][k] < 0) continue;
            const int idx = off[a][k];
            const int* crop = p.crop[a];
            if (crop[1] > crop[0]) {
                const int hh = idx / p.W, ww = idx - hh * p.W;
                if (hh < crop[0] || hh >= crop[1] || ww < crop[2] || ww >= crop[3]) continue;
            }
            const float sg = 1.f / (1.f + expf(-occ[(size_t)a * HW + idx]));
            unsafeAtomicAdd(gocc + (size_t)a * HW + idx, dS * tw[a][k] * sg * (1.f - sg));
        }
    }
}

static int fill_params(WarpParams& p, int n_agents, int C, int H, int W, const double* affine_host,
                       const double* affine_dev, int grid_f64, const int32_t* crop_host) {
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= WF_MAXA, "warp_fuse: n_agents must be in [1,%d] (got %d)",
                 WF_MAXA, n_agents);
    HEAL_REQUIRE(C >= 1 && H >= 1 && W >= 1, "warp_fuse: bad shape");
    HEAL_REQUIRE(affine_host != nullptr || affine_dev != nullptr, "warp_fuse: affine is NULL (host and device)");
    p.n_agents = n_agents; p.C = C; p.H = H; p.W = W; p.grid_f64 = grid_f64;
    p.mdev = affine_dev;
    for (int a = 0; a < WF_MAXA; ++a) {
        for (int k = 0; k < 6; ++k) p.m[a][k] = (a < n_agents && affine_host) ? affine_host[a * 6 + k] : 0.0;
        for (int k = 0; k < 4; ++k) p.crop[a][k] = (a < n_agents && crop_host) ? crop_host[a * 4 + k] : 0;
    }
    return 0;
}

}  // namespace heal

using namespace heal;

extern "C" int heal_warp_fuse(const float* feats, const float* occ, int n_agents, int channels, int H,
                              int W, const double* affine_host, const double* affine_dev, int grid_f64,
                              const int32_t* crop_host, float* out, void* stream) {
    WarpParams p;
    if (fill_params(p, n_agents, channels, H, W, affine_host, affine_dev, grid_f64, crop_host)) return 1;
    // channels per block: the per-pixel prologue (grid, taps, scores, softmax) is recomputed by every
    // channel block, so use as few channel blocks as still give ~1024 workgroups
    const int tiles = ceil_div(W, WF_TW) * ceil_div(H, WF_TH);
    int cch = (int)(((long long)channels * tiles + 1023) / 1024);
    cch = (cch + 3) / 4 * 4;
    if (cch < 4) cch = 4;
    if (cch > channels) cch = (channels + 3) / 4 * 4;
    dim3 grid(ceil_div(W, WF_TW), ceil_div(H, WF_TH), ceil_div(channels, cch));
    hipStream_t st = (hipStream_t)stream;
    switch (n_agents) {
#define HEAL_WF(N) case N: k_warp_fuse<N><<<grid, 256, 0, st>>>(feats, occ, p, cch, out); break;
        HEAL_WF(1) HEAL_WF(2) HEAL_WF(3) HEAL_WF(4) HEAL_WF(5) HEAL_WF(6) HEAL_WF(7) HEAL_WF(8)
#undef HEAL_WF
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_warp_fuse_levels(int n_levels, const float* const* feats_host, const float* const* occ_host,
                                     int n_agents, const int32_t* channels_host, const int32_t* h_host, const int32_t* w_host,
                                     const double* affine_host, const double* affine_dev, int grid_f64,
                                     const int32_t* crop_host, float* const* out_host, void* stream) {
    HEAL_REQUIRE(n_levels >= 1 && n_levels <= WL_MAXL, "warp_fuse_levels: 1..%d levels per launch (got %d)", WL_MAXL, n_levels);
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= WF_MAXA, "warp_fuse_levels: n_agents must be in [1,%d] (got %d)", WF_MAXA, n_agents);
    HEAL_REQUIRE(affine_host != nullptr || affine_dev != nullptr, "warp_fuse_levels: affine is NULL (host and device)");
    WfLevels P;
    P.n_levels = n_levels; P.n_agents = n_agents; P.grid_f64 = grid_f64; P.mdev = affine_dev;
    P.dbg = HEAL_DEBUG_ENV("HEAL_K5_DBG");   // timing experiments only (read once, announced on stderr)
    for (int a = 0; a < WF_MAXA; ++a)
        for (int k = 0; k < 6; ++k) P.m[a][k] = (a < n_agents && affine_host) ? affine_host[a * 6 + k] : 0.0;
    long long blocks = 0;
    for (int l = 0; l < WL_MAXL; ++l) {
        WfLevel& L = P.lv[l];
        if (l >= n_levels) { L = P.lv[0]; L.block0 = 1 << 30; continue; }
        HEAL_REQUIRE(feats_host[l] && occ_host[l] && out_host[l] && channels_host[l] >= 1 && h_host[l] >= 1 && w_host[l] >= 1,
                     "warp_fuse_levels: bad level %d", l);
        L.feats = feats_host[l]; L.occ = occ_host[l]; L.out = out_host[l];
        L.C = channels_host[l]; L.H = h_host[l]; L.W = w_host[l];
        L.tiles_x = ceil_div(L.W, WL_T);
        const int tiles = L.tiles_x * ceil_div(L.H, WL_T);
        // channel slices per tile: the per-pixel prologue (fp64 grid, taps, 4 n_agents score samples with their sigmoids, softmax:
        // half of the kernel's time at 2048 blocks per level) is recomputed by every slice, so as few slices as still give ~512
        // blocks per level, in whole staging rounds of 8 channels.  Measured at scene5 size (scripts/k5_bench.py, HEAL_K5_BLOCKS):
        // 2048 -> 110 us, 1024 -> 102, 768 -> 91, 512 -> 89, 384 -> 97, 256 -> 116 (three per-level launches of round 3: 135).
        int target = 512;
        static const int env_blocks = [] { const char* e = getenv("HEAL_K5_BLOCKS"); return e ? atoi(e) : 0; }();   // tuning knob, read once
        if (env_blocks > 0) target = env_blocks;
        int cg = (int)((target + tiles - 1) / tiles);
        cg = cg < 1 ? 1 : cg;
        int cpb = ceil_div(ceil_div(L.C, cg), WL_CC) * WL_CC;
        L.cpb = cpb;
        L.cgroups = ceil_div(L.C, cpb);
        L.block0 = (int)blocks;
        blocks += (long long)tiles * L.cgroups;
        for (int a = 0; a < WF_MAXA; ++a)
            for (int k = 0; k < 4; ++k)
                L.crop[a][k] = (a < n_agents && crop_host) ? crop_host[((size_t)l * n_agents + a) * 4 + k] : 0;
    }
    HEAL_REQUIRE(blocks < (1ll << 30), "warp_fuse_levels: grid too large");
    hipStream_t st = (hipStream_t)stream;
    switch (n_agents) {
#define HEAL_WFL(N) case N: HEAL_LAUNCH_EV(k_warp_fuse_lds<N>, dim3((unsigned)blocks), dim3(256), 0, st, P); break;
        HEAL_WFL(1) HEAL_WFL(2) HEAL_WFL(3) HEAL_WFL(4) HEAL_WFL(5) HEAL_WFL(6) HEAL_WFL(7) HEAL_WFL(8)
#undef HEAL_WFL
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_warp_agent(const float* feat, const float* occ, int channels, int H, int W,
                               const double* affine_host, const double* affine_dev, int grid_f64,
                               const int32_t* crop_host, float* feat_ego, float* score_ego, void* stream) {
    WarpParams p;
    if (fill_params(p, 1, channels, H, W, affine_host, affine_dev, grid_f64, crop_host)) return 1;
    constexpr int CCH = 16;
    dim3 grid(ceil_div(W, WF_TW), ceil_div(H, WF_TH), ceil_div(channels, CCH));
    k_warp_agent<CCH><<<grid, 256, 0, (hipStream_t)stream>>>(feat, occ, p, feat_ego, score_ego);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_warp_agents_pm(const float* feats, int n_agents, int channels, int H, int W,
                                   const double* affine_host, const double* affine_dev, int grid_f64, float* out,
                                   void* stream) {
    WarpParams p;
    if (fill_params(p, n_agents, channels, H, W, affine_host, affine_dev, grid_f64, nullptr)) return 1;
    HEAL_REQUIRE(feats && out, "warp_agents_pm: null pointer");
    HEAL_REQUIRE(channels % 4 == 0, "warp_agents_pm: channels must be a multiple of 4 (got %d)", channels);
    dim3 grid(ceil_div(W, WPM_TW), ceil_div(H, WPM_TH), n_agents * ceil_div(channels, WPM_C));
    k_warp_agents_pm<<<grid, 256, 0, (hipStream_t)stream>>>(feats, p, out);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_fuse_warped(const float* feats_ego, const float* scores_ego, int n_agents,
                                int channels, int H, int W, float* out, void* stream) {
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= WF_MAXA, "fuse_warped: n_agents must be in [1,%d]", WF_MAXA);
    HEAL_REQUIRE((H * W) % 4 == 0, "fuse_warped: H*W must be a multiple of 4");
    const int HW4 = H * W / 4;
    FuseOffsets off;
    for (int a = 0; a < WF_MAXA; ++a) {
        off.f[a] = (long long)a * channels * HW4;
        off.s[a] = (long long)a * HW4;
    }
    dim3 grid(ceil_div(HW4, 256), ceil_div(channels, 16));
    k_fuse_warped<<<grid, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(feats_ego), reinterpret_cast<const float4*>(scores_ego), off, n_agents,
        channels, HW4, reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_fuse_warped_rows(const float* base, const int64_t* feat_offsets_host, const int64_t* score_offsets_host,
                                     int n_agents, int channels, int H, int W, float* out, void* stream) {
    HEAL_REQUIRE(n_agents >= 1 && n_agents <= WF_MAXA, "fuse_warped_rows: n_agents must be in [1,%d]", WF_MAXA);
    HEAL_REQUIRE((H * W) % 4 == 0 && ((uintptr_t)base & 15) == 0, "fuse_warped_rows: H*W %% 4 == 0 and a 16-B aligned base");
    const int HW4 = H * W / 4;
    FuseOffsets off;
    for (int a = 0; a < WF_MAXA; ++a) {
        const int64_t fo = a < n_agents ? feat_offsets_host[a] : 0, so = a < n_agents ? score_offsets_host[a] : 0;
        HEAL_REQUIRE(fo % 4 == 0 && so % 4 == 0 && fo >= 0 && so >= 0, "fuse_warped_rows: offsets must be multiples of 4 floats");
        off.f[a] = fo / 4;
        off.s[a] = so / 4;
    }
    dim3 grid(ceil_div(HW4, 256), ceil_div(channels, 16));
    k_fuse_warped<<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const float4*>(base),
                                                         reinterpret_cast<const float4*>(base), off, n_agents, channels, HW4,
                                                         reinterpret_cast<float4*>(out));
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_warp_fuse_backward(const float* feats, const float* occ, int n_agents, int channels, int H, int W,
                                       const double* affine_host, const double* affine_dev, int grid_f64,
                                       const int32_t* crop_host, const float* grad_out, float* grad_feats, float* grad_occ,
                                       void* stream) {
    WarpParams p;
    if (fill_params(p, n_agents, channels, H, W, affine_host, affine_dev, grid_f64, crop_host)) return 1;
    HEAL_REQUIRE(feats && occ && grad_out && grad_feats && grad_occ, "warp_fuse_backward: null pointer");
    dim3 grid(ceil_div(W, WF_TW), ceil_div(H, WF_TH));
    hipStream_t st = (hipStream_t)stream;
    switch (n_agents) {
#define HEAL_WFB(N) case N: k_warp_fuse_backward<N><<<grid, 256, 0, st>>>(feats, occ, p, grad_out, grad_feats, grad_occ); break;
        HEAL_WFB(1) HEAL_WFB(2) HEAL_WFB(3) HEAL_WFB(4) HEAL_WFB(5) HEAL_WFB(6) HEAL_WFB(7) HEAL_WFB(8)
#undef HEAL_WFB
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}
