// K2 -- fused pillar feature net (PillarVFE + PFNLayer) and scatter to the dense BEV canvas.
//
// Reference arithmetic: opencood/models/sub_modules/pillar_vfe.py:105-155 (feature decoration and
// padding mask), :31-53 (Linear(10->64, no bias) -> eval BatchNorm1d(eps 1e-3) -> ReLU -> max over
// the P rows INCLUDING the zeroed padding rows), opencood/models/sub_modules/point_pillar_scatter.py:
// 19-76 (canvas[:, z + y*nx + x] = pillar, one zero canvas per agent).
//
// MI355X formulation
//   k_pfn     one 64-lane wave per pillar, lane = output channel.  The P x 10 decorated features
//             are staged in LDS and read back as broadcasts, each lane runs the 10-term dot product
//             for its channel, BN scale/shift, ReLU and the running max.  The pillar's canvas cell
//             is recorded in an index map (cell -> pillar id).
//   k_canvas  one pass over the WHOLE canvas: each thread owns 4 consecutive x cells, reads their
//             pillar ids once and streams 16-B stores for every channel (zeros where the cell is
//             empty).  The 67 MB/agent canvas is therefore written exactly once -- no memset pass
//             followed by a scatter pass -- which is what the HBM roofline of this operator allows.
#include <stdlib.h>
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

constexpr int PFN_C = 64;
constexpr int PFN_FROW = 12;  // 10 features padded to 12 floats (16-B aligned rows)

struct PfnGeom {
    float vx, vy, vz, xo, yo, zo;
    int n_agents, ny, nx;
};

__global__ __launch_bounds__(256) void k_pfn(const float4* __restrict__ voxels, int P,
                                            const int4* __restrict__ coords,
                                            const int* __restrict__ num_points, int n_voxels,
                                            const int* __restrict__ n_voxels_dev,
                                            const float* __restrict__ weight /*[64][10]*/,
                                            const float* __restrict__ bn_scale,
                                            const float* __restrict__ bn_shift, PfnGeom g,
                                            float* __restrict__ pillar_feat /*[M][64]*/,
                                            int* __restrict__ cell_map /*[n_agents][ny*nx]*/) {
    __shared__ __attribute__((aligned(16))) float sfeat[4][64][PFN_FROW];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    int M = n_voxels;
    if (n_voxels_dev != nullptr) M = min(M, *n_voxels_dev);

    float w[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = weight[l * 10 + k];
    const float sc = bn_scale[l], sh = bn_shift[l];
    const float pad_val = fmaxf(sh, 0.f);  // a zeroed row gives 0*W -> BN -> ReLU = relu(shift)

    const int stride = gridDim.x * 4;
    int m = blockIdx.x * 4 + wave;
    int4 cd_n = make_int4(0, 0, 0, 0);
    int np_n = 0;
    float4 pt_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) {
        cd_n = coords[m]; np_n = num_points[m];
        if (l < P) pt_n = voxels[(size_t)m * P + l];
    }
    for (; m < M; m += stride) {
        const int4 cd = cd_n;  // (b, z, y, x)
        const int np = np_n;
        const float4 pt = pt_n;
        // prefetch the next pillar of this wave while this one is being reduced
        const int mn = m + stride;
        if (mn < M) {
            cd_n = coords[mn]; np_n = num_points[mn];
            pt_n = (l < P) ? voxels[(size_t)mn * P + l] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // mean over the COUNT of the sum over all P rows (pillar_vfe.py:118-121)
        const float fn = (float)np;
        const float mx = wave_sum(pt.x) / fn;
        const float my = wave_sum(pt.y) / fn;
        const float mz = wave_sum(pt.z) / fn;
        if (l < P) {
            const bool live = l < np;
            const float cxm = (float)cd.w * g.vx + g.xo;
            const float cym = (float)cd.z * g.vy + g.yo;
            const float czm = (float)cd.y * g.vz + g.zo;
            float f[PFN_FROW];
            f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w;
            f[4] = pt.x - mx; f[5] = pt.y - my; f[6] = pt.z - mz;
            f[7] = pt.x - cxm; f[8] = pt.y - cym; f[9] = pt.z - czm;
            f[10] = 0.f; f[11] = 0.f;
            float4* dst = reinterpret_cast<float4*>(&sfeat[wave][l][0]);
            const float k = live ? 1.f : 0.f;  // features *= mask (pillar_vfe.py:145-149)
            dst[0] = make_float4(f[0] * k, f[1] * k, f[2] * k, f[3] * k);
            dst[1] = make_float4(f[4] * k, f[5] * k, f[6] * k, f[7] * k);
            dst[2] = make_float4(f[8] * k, f[9] * k, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float best = (np < P) ? pad_val : 0.f;  // ReLU output is >= 0, so 0 is the identity of max
        const int live_n = min(np, P);
        for (int p = 0; p < live_n; ++p) {
            const float4* src = reinterpret_cast<const float4*>(&sfeat[wave][p][0]);
            const float4 a = src[0], b = src[1], c = src[2];
            float acc = a.x * w[0];
            acc = fmaf(a.y, w[1], acc); acc = fmaf(a.z, w[2], acc); acc = fmaf(a.w, w[3], acc);
            acc = fmaf(b.x, w[4], acc); acc = fmaf(b.y, w[5], acc); acc = fmaf(b.z, w[6], acc);
            acc = fmaf(b.w, w[7], acc); acc = fmaf(c.x, w[8], acc); acc = fmaf(c.y, w[9], acc);
            const float y = fmaf(acc, sc, sh);
            best = fmaxf(best, y);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pillar_feat[(size_t)m * PFN_C + l] = best;
        if (l == 0) {
            const int idx = cd.y + cd.z * g.nx + cd.w;  // z + y*nx + x (point_pillar_scatter.py:58)
            if (cd.x >= 0 && cd.x < g.n_agents && idx >= 0 && idx < g.ny * g.nx)
                atomicMax(&cell_map[(size_t)cd.x * g.ny * g.nx + idx], m);
        }
    }
}

// Four pillars per wave: 16 lanes per pillar, 4 output channels per lane.  The one-pillar-per-wave kernel above is
// load-latency-bound (a pillar is ~5 points of work); packing four pillars into a wave quarters the number of waves that
// each wait out the same round trip (33 k pillars: 24.6 -> see DESIGN.md), and a lane's 4 channels leave as one 16-B store.
// P <= 32 (two points per lane).  Same arithmetic per channel as k_pfn: 10-term fmaf chain, BN, ReLU, running max.
__global__ __launch_bounds__(256) void k_pfn4(const float4* __restrict__ voxels, int P,
                                             const int4* __restrict__ coords,
                                             const int* __restrict__ num_points, int n_voxels,
                                             const int* __restrict__ n_voxels_dev,
                                             const float* __restrict__ weight /*[64][10]*/,
                                             const float* __restrict__ bn_scale,
                                             const float* __restrict__ bn_shift, PfnGeom g,
                                             float* __restrict__ pillar_feat /*[M][64]*/,
                                             int* __restrict__ cell_map /*[n_agents][ny*nx]*/) {
    __shared__ __attribute__((aligned(16))) float sfeat[4][4][32][PFN_FROW];  // [wave][pillar][point][feature]
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int q = l >> 4, i = l & 15;  // pillar slot of the wave, lane inside the 16-lane group
    int M = n_voxels;
    if (n_voxels_dev != nullptr) M = min(M, *n_voxels_dev);
    float w[4][10], sc[4], sh[4], padv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = i * 4 + j;
#pragma unroll
        for (int k = 0; k < 10; ++k) w[j][k] = weight[c * 10 + k];
        sc[j] = bn_scale[c]; sh[j] = bn_shift[c];
        padv[j] = fmaxf(sh[j], 0.f);
    }
    const int stride = gridDim.x * 16;
    for (int m0 = (blockIdx.x * 4 + wave) * 4; m0 < M; m0 += stride) {
        const int m = m0 + q;
        const bool live_p = m < M;
        int4 cd = make_int4(0, 0, 0, 0);
        int np = 0;
        float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
        if (live_p) {
            cd = coords[m]; np = num_points[m];
            if (i < P) pa = voxels[(size_t)m * P + i];
            if (i + 16 < P) pb = voxels[(size_t)m * P + i + 16];
        }
        // mean over the COUNT of the sum over all P rows (pillar_vfe.py:118-121): reduce inside the 16-lane group
        float sx = pa.x + pb.x, sy = pa.y + pb.y, sz = pa.z + pb.z;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64);
        }
        const float fn = (float)np;
        const float mx = sx / fn, my = sy / fn, mz = sz / fn;
        const float cxm = (float)cd.w * g.vx + g.xo, cym = (float)cd.z * g.vy + g.yo, czm = (float)cd.y * g.vz + g.zo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = i + 16 * h;
            const float4 pt = h ? pb : pa;
            const float k = p < np ? 1.f : 0.f;  // features *= mask (pillar_vfe.py:145-149)
            float4* dst = reinterpret_cast<float4*>(&sfeat[wave][q][p][0]);
            dst[0] = make_float4(pt.x * k, pt.y * k, pt.z * k, pt.w * k);
            dst[1] = make_float4((pt.x - mx) * k, (pt.y - my) * k, (pt.z - mz) * k, (pt.x - cxm) * k);
            dst[2] = make_float4((pt.y - cym) * k, (pt.z - czm) * k, 0.f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float best[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) best[j] = (np < P) ? padv[j] : 0.f;  // zeroed rows give relu(shift); ReLU >= 0
        const int live_n = min(np, P);
        for (int p = 0; p < live_n; ++p) {
            const float4* src = reinterpret_cast<const float4*>(&sfeat[wave][q][p][0]);
            const float4 a = src[0], b = src[1], c = src[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float acc = a.x * w[j][0];
                acc = fmaf(a.y, w[j][1], acc); acc = fmaf(a.z, w[j][2], acc); acc = fmaf(a.w, w[j][3], acc);
                acc = fmaf(b.x, w[j][4], acc); acc = fmaf(b.y, w[j][5], acc); acc = fmaf(b.z, w[j][6], acc);
                acc = fmaf(b.w, w[j][7], acc); acc = fmaf(c.x, w[j][8], acc); acc = fmaf(c.y, w[j][9], acc);
                best[j] = fmaxf(best[j], fmaf(acc, sc[j], sh[j]));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (live_p) {
            *reinterpret_cast<float4*>(&pillar_feat[(size_t)m * PFN_C + i * 4]) = make_float4(best[0], best[1], best[2], best[3]);
            if (i == 0) {
                const int idx = cd.y + cd.z * g.nx + cd.w;  // z + y*nx + x (point_pillar_scatter.py:58)
                if (cell_map && cd.x >= 0 && cd.x < g.n_agents && idx >= 0 && idx < g.ny * g.nx)
                    atomicMax(&cell_map[(size_t)cd.x * g.ny * g.nx + idx], m);
            }
        }
    }
}

// ---- training: batch statistics and the backward of Linear(10 -> 64, no bias) -> BatchNorm1d -> ReLU -> max ----
// pillar_vfe.py:25-51 (PFNLayer) in training mode normalises with the statistics of ALL M x P rows (zeroed padding rows
// included).  With z = W f the per-channel moments are closed forms of two small sums over the rows,
//     s1 = sum_rows f [10],   S = sum_rows f f^T [10 x 10]:   mean_c = W_c . s1 / R,   E[z_c^2] = W_c^T S W_c / R,
// and so is every row-sum the weight gradient needs:  sum_rows xhat_c f = rstd_c ((W S)_c - mean_c s1).
// k_pfn_moments: per-block partial sums of the 10 + 55 distinct products (the host adds the partials in float64).
// k_pfn_backward: per pillar and channel the arg-max row p* of y = relu(scale z + shift) (a padding row counts once: z = 0,
//   f = 0), dy = g[m][c] where y* > 0; per-block partial sums of
//     A[c][k] = sum dy f_{p*}[k]  (64 x 10),   B[c] = sum dy,   Cx[c] = sum dy xhat_{p*}[c],   xhat = (z - mean) rstd.
//   dW = (gamma rstd) (A - (B/R) s1 - (Cx/R) rstd ((W S) - mean s1))   [batch statistics]   |   scale A   [running statistics]
//   dgamma = Cx, dbeta = B.
// Both kernels stage the decorated features exactly as k_pfn4 does (same arithmetic: the arg-max agrees with the forward).
constexpr int PFN_NMOM = 65;    // 10 sums + 55 products f_j f_k (j <= k)
constexpr int PFN_BWD_ROW = 12; // A[c][0..10), B[c], Cx[c]

__device__ __forceinline__ void pfn_stage4(const float4* __restrict__ voxels, int P, const int4* __restrict__ coords,
                                           const int* __restrict__ num_points, int M, int m, int i, const PfnGeom& g,
                                           float (*sf)[PFN_FROW] /*[32][12] of this pillar slot*/, int& np_out) {
    const bool live_p = m < M;
    int4 cd = make_int4(0, 0, 0, 0);
    int np = 0;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
    if (live_p) {
        cd = coords[m]; np = num_points[m];
        if (i < P) pa = voxels[(size_t)m * P + i];
        if (i + 16 < P) pb = voxels[(size_t)m * P + i + 16];
    }
    float sx = pa.x + pb.x, sy = pa.y + pb.y, sz = pa.z + pb.z;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64);
    }
    const float fn = (float)np;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    const float cxm = (float)cd.w * g.vx + g.xo, cym = (float)cd.z * g.vy + g.yo, czm = (float)cd.y * g.vz + g.zo;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int p = i + 16 * h;
        const float4 pt = h ? pb : pa;
        const float k = p < np ? 1.f : 0.f;
        float4* dst = reinterpret_cast<float4*>(&sf[p][0]);
        dst[0] = make_float4(pt.x * k, pt.y * k, pt.z * k, pt.w * k);
        dst[1] = make_float4((pt.x - mx) * k, (pt.y - my) * k, (pt.z - mz) * k, (pt.x - cxm) * k);
        dst[2] = make_float4((pt.y - cym) * k, (pt.z - czm) * k, 0.f, 0.f);
    }
    np_out = live_p ? np : 0;
}

__global__ __launch_bounds__(256) void k_pfn_moments(const float4* __restrict__ voxels, int P, const int4* __restrict__ coords,
                                                    const int* __restrict__ num_points, int M, PfnGeom g,
                                                    float* __restrict__ partials /*[gridDim.x][65]*/) {
    __shared__ __attribute__((aligned(16))) float sfeat[4][4][32][PFN_FROW];
    __shared__ float s_red[4][PFN_NMOM];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, i = l & 15;
    float acc[PFN_NMOM];
#pragma unroll
    for (int k = 0; k < PFN_NMOM; ++k) acc[k] = 0.f;
    const int stride = gridDim.x * 16;
    for (int m0 = (blockIdx.x * 4 + wave) * 4; m0 < M; m0 += stride) {
        int np;
        pfn_stage4(voxels, P, coords, num_points, M, m0 + q, i, g, sfeat[wave][q], np);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p = i + 16 * h;
            if (p >= min(np, P)) continue;       // padding rows are zero rows: nothing to add
            float f[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) f[k] = sfeat[wave][q][p][k];
            int t = 10;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                acc[j] += f[j];
#pragma unroll
                for (int k = j; k < 10; ++k) { acc[t] = fmaf(f[j], f[k], acc[t]); ++t; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < PFN_NMOM; ++k) {
        const float v = wave_sum(acc[k]);
        if (l == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < PFN_NMOM)
        partials[(size_t)blockIdx.x * PFN_NMOM + threadIdx.x] =
            (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_pfn_backward(const float4* __restrict__ voxels, int P, const int4* __restrict__ coords,
                                                     const int* __restrict__ num_points, int M,
                                                     const float* __restrict__ weight /*[64][10]*/,
                                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, PfnGeom g,
                                                     const float* __restrict__ grad /*[M][64]*/,
                                                     float* __restrict__ partials /*[gridDim.x][64][12]*/) {
    __shared__ __attribute__((aligned(16))) float sfeat[4][4][32][PFN_FROW];
    __shared__ float s_red[4][16][4 * PFN_BWD_ROW];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, q = l >> 4, i = l & 15;
    float w[4][10], sc[4], sh[4], mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = i * 4 + j;
#pragma unroll
        for (int k = 0; k < 10; ++k) w[j][k] = weight[c * 10 + k];
        sc[j] = bn_scale[c]; sh[j] = bn_shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
    }
    float acc[4][PFN_BWD_ROW];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < PFN_BWD_ROW; ++k) acc[j][k] = 0.f;
    const int stride = gridDim.x * 16;
    for (int m0 = (blockIdx.x * 4 + wave) * 4; m0 < M; m0 += stride) {
        const int m = m0 + q;
        int np;
        pfn_stage4(voxels, P, coords, num_points, M, m, i, g, sfeat[wave][q], np);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (m < M) {
            const int live_n = min(np, P);
            float best[4], zb[4];
            int pb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {      // the padding row (if any): z = 0, y = relu(shift), no features
                best[j] = (np < P) ? fmaxf(sh[j], 0.f) : -1.f;   // (y >= 0: -1 loses against every row)
                zb[j] = 0.f; pb[j] = -1;
            }
            for (int p = 0; p < live_n; ++p) {
                const float4* src = reinterpret_cast<const float4*>(&sfeat[wave][q][p][0]);
                const float4 a = src[0], b = src[1], c = src[2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float z = a.x * w[j][0];
                    z = fmaf(a.y, w[j][1], z); z = fmaf(a.z, w[j][2], z); z = fmaf(a.w, w[j][3], z);
                    z = fmaf(b.x, w[j][4], z); z = fmaf(b.y, w[j][5], z); z = fmaf(b.z, w[j][6], z);
                    z = fmaf(b.w, w[j][7], z); z = fmaf(c.x, w[j][8], z); z = fmaf(c.y, w[j][9], z);
                    const float y = fmaxf(fmaf(z, sc[j], sh[j]), 0.f);
                    if (y > best[j]) { best[j] = y; zb[j] = z; pb[j] = p; }
                }
            }
            const float4 gv = *reinterpret_cast<const float4*>(&grad[(size_t)m * PFN_C + i * 4]);
            const float gj[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = best[j] > 0.f ? gj[j] : 0.f;
                acc[j][10] += dy;
                acc[j][11] = fmaf(dy, (zb[j] - mu[j]) * rs[j], acc[j][11]);
                if (pb[j] >= 0) {
#pragma unroll
                    for (int k = 0; k < 10; ++k) acc[j][k] = fmaf(dy, sfeat[wave][q][pb[j]][k], acc[j][k]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // add the four pillar slots of the wave (lanes i, i + 16, i + 32, i + 48 own the same channels), then the four waves
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < PFN_BWD_ROW; ++k) {
            float v = acc[j][k];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) s_red[wave][i][j * PFN_BWD_ROW + k] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 4 * PFN_BWD_ROW; e += 256) {
        const int ii = e / (4 * PFN_BWD_ROW), r = e - ii * (4 * PFN_BWD_ROW);
        partials[(size_t)blockIdx.x * (PFN_C * PFN_BWD_ROW) + e] =
            (s_red[0][ii][r] + s_red[1][ii][r]) + (s_red[2][ii][r] + s_red[3][ii][r]);
    }
}

using vf4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void st_nt(float4* p, float4 v) {
    vf4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
}

// canvas[b][c][cell] = cell_map[b][cell] >= 0 ? pillar_feat[id][c] : 0, 4 cells per thread
template <int CG /*channels per block*/, bool NT>
__global__ __launch_bounds__(256) void k_canvas(const int4* __restrict__ cell_map4,
                                               const float* __restrict__ pillar_feat, int cells4,
                                               int channels, float4* __restrict__ canvas4) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= cells4) return;
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CG;
    const int4 id = cell_map4[(size_t)b * cells4 + t];
    float4* out = canvas4 + ((size_t)b * channels + c0) * cells4 + t;
    if ((id.x & id.y & id.z & id.w) < 0) {  // all four empty (ids are -1)
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            if (NT) st_nt(&out[(size_t)c * cells4], z);
            else out[(size_t)c * cells4] = z;
        }
        return;
    }
#pragma unroll 4
    for (int c = 0; c < CG; ++c) {
        float4 v;
        v.x = id.x >= 0 ? pillar_feat[(size_t)id.x * channels + c0 + c] : 0.f;
        v.y = id.y >= 0 ? pillar_feat[(size_t)id.y * channels + c0 + c] : 0.f;
        v.z = id.z >= 0 ? pillar_feat[(size_t)id.z * channels + c0 + c] : 0.f;
        v.w = id.w >= 0 ? pillar_feat[(size_t)id.w * channels + c0 + c] : 0.f;
        if (NT) st_nt(&out[(size_t)c * cells4], v);
        else out[(size_t)c * cells4] = v;
    }
}

}  // namespace heal

using namespace heal;

// Shared with the BEV pool (K4): stream a dense [n,C,cells] canvas from a cell->row index map.
int heal_canvas_from_map(const int* cell_map, const float* rows, int n_agents, int channels,
                         int cells, float* canvas, hipStream_t s) {
    HEAL_REQUIRE(cells % 4 == 0 && channels % 16 == 0, "canvas: cells %% 4 and channels %% 16 required");
    const int cells4 = cells / 4;
    // channels per block: measured on the 201 MB collated canvas (scripts/k2_bench.py): 1: 77 us, 2: 72, 4: 75, 8: 79,
    // 16: 86, 32: 98, 64: 93 -> many small blocks win (more stores in flight); HEAL_CANVAS_CG overrides for tuning
    static const int cg = []() { const char* e = getenv("HEAL_CANVAS_CG"); return e ? atoi(e) : 4; }();
    static const int nt = []() { const char* e = getenv("HEAL_CANVAS_NT"); return e ? atoi(e) : 0; }();
    const int4* map4 = reinterpret_cast<const int4*>(cell_map);
    float4* out4 = reinterpret_cast<float4*>(canvas);
#define HEAL_CV(CG_)                                                                                         \
    if (cg == CG_ && channels % CG_ == 0) {                                                                  \
        dim3 grid(ceil_div(cells4, 256), channels / CG_, n_agents);                                          \
        if (nt) k_canvas<CG_, true><<<grid, 256, 0, s>>>(map4, rows, cells4, channels, out4);                \
        else k_canvas<CG_, false><<<grid, 256, 0, s>>>(map4, rows, cells4, channels, out4);                  \
        HEAL_LAUNCH_CHECK();                                                                                 \
        return 0;                                                                                            \
    }
    HEAL_CV(1) HEAL_CV(2) HEAL_CV(4) HEAL_CV(8) HEAL_CV(32) HEAL_CV(64)
    // 16 (or a channel count the choice does not divide)
#undef HEAL_CV
    dim3 grid(ceil_div(cells4, 256), channels / 16, n_agents);
    if (nt) k_canvas<16, true><<<grid, 256, 0, s>>>(map4, rows, cells4, channels, out4);
    else k_canvas<16, false><<<grid, 256, 0, s>>>(map4, rows, cells4, channels, out4);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_pfn_scatter_workspace(int n_voxels, int n_agents, int ny, int nx, int channels) {
    size_t b = align_up((size_t)n_agents * ny * nx * sizeof(int));
    b += align_up((size_t)(n_voxels < 1 ? 1 : n_voxels) * channels * sizeof(float));
    return b + 256;
}

// PillarVFE + PFN for the collated agents of a modality: pillar features [M, 64] and the cell -> pillar-row map (-1 = empty cell)
static int pfn_pillars_impl(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                            const int32_t* n_voxels_dev, int max_points, const float* weight, const float* bn_scale,
                            const float* bn_shift, int channels, float vx, float vy, float vz, float x_offset, float y_offset,
                            float z_offset, int n_agents, int ny, int nx, float* pf, int* cell_map, hipStream_t s,
                            const char* who) {
    HEAL_REQUIRE(channels == PFN_C, "%s: channels must be 64 (got %d)", who, channels);
    HEAL_REQUIRE(max_points >= 1 && max_points <= 64, "%s: max_points must be in [1,64]", who);
    HEAL_REQUIRE(n_agents >= 1 && ny >= 1 && nx >= 1 && (nx * ny) % 4 == 0, "%s: bad grid", who);
    HEAL_REQUIRE(n_voxels >= 0, "%s: negative n_voxels", who);
    HEAL_FILL(cell_map, 0xFF, (size_t)n_agents * ny * nx * sizeof(int), s);
    if (n_voxels > 0) {
        PfnGeom g{vx, vy, vz, x_offset, y_offset, z_offset, n_agents, ny, nx};
        // one pillar per wave up to 32768 waves (a collated 3-agent launch is ~33 k pillars), beyond that a
        // (software-prefetched) grid-stride loop; measured: fewer, longer-lived waves are slower (the per-pillar
        // work is ~5 points on average)
        const int blocks = min(ceil_div(n_voxels, 4), 256 * 32);
        const float4* v4 = reinterpret_cast<const float4*>(voxels);
        const int4* c4 = reinterpret_cast<const int4*>(coords);
        static const bool one_per_wave = getenv("HEAL_PFN_1PW") != nullptr;  // A/B switch
        if (max_points <= 32 && !one_per_wave)
            k_pfn4<<<min(ceil_div(n_voxels, 16), 256 * 16), 256, 0, s>>>(v4, max_points, c4, num_points, n_voxels,
                                                                         n_voxels_dev, weight, bn_scale, bn_shift, g, pf,
                                                                         cell_map);
        else
            k_pfn<<<blocks, 256, 0, s>>>(v4, max_points, c4, num_points, n_voxels, n_voxels_dev, weight,
                                         bn_scale, bn_shift, g, pf, cell_map);
        HEAL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int heal_pfn_scatter(const float* voxels, const int32_t* coords, const int32_t* num_points,
                                int n_voxels, const int32_t* n_voxels_dev, int max_points,
                                const float* weight, const float* bn_scale, const float* bn_shift,
                                int channels, float vx, float vy, float vz, float x_offset,
                                float y_offset, float z_offset, int n_agents, int ny, int nx,
                                float* canvas, float* pillar_feat, void* ws, size_t ws_bytes,
                                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && ny >= 1 && nx >= 1 && n_voxels >= 0, "pfn_scatter: bad sizes");
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "pfn_scatter: workspace must be 256-B aligned");
    Arena a(ws, ws_bytes);
    int* cell_map = a.take<int>((size_t)n_agents * ny * nx);
    float* pf = pillar_feat;
    if (pf == nullptr) pf = a.take<float>((size_t)(n_voxels < 1 ? 1 : n_voxels) * channels);
    HEAL_REQUIRE(a.ok(), "pfn_scatter: workspace too small (%zu < %zu)", ws_bytes, a.off);
    if (pfn_pillars_impl(voxels, coords, num_points, n_voxels, n_voxels_dev, max_points, weight, bn_scale, bn_shift, channels, vx,
                         vy, vz, x_offset, y_offset, z_offset, n_agents, ny, nx, pf, cell_map, s, "pfn_scatter"))
        return 1;
    return heal_canvas_from_map(cell_map, pf, n_agents, channels, ny * nx, canvas, s);
}

extern "C" int heal_pfn_pillars(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                                const int32_t* n_voxels_dev, int max_points, const float* weight, const float* bn_scale,
                                const float* bn_shift, int channels, float vx, float vy, float vz, float x_offset,
                                float y_offset, float z_offset, int n_agents, int ny, int nx, float* pillar_feat,
                                int32_t* cell_map, void* stream) {
    HEAL_REQUIRE(pillar_feat && cell_map && ((uintptr_t)cell_map & 15) == 0, "pfn_pillars: null / misaligned output");
    return pfn_pillars_impl(voxels, coords, num_points, n_voxels, n_voxels_dev, max_points, weight, bn_scale, bn_shift, channels,
                            vx, vy, vz, x_offset, y_offset, z_offset, n_agents, ny, nx, pillar_feat, cell_map,
                            (hipStream_t)stream, "pfn_pillars");
}

extern "C" int heal_pillar_canvas(const int32_t* cell_map, const float* pillar_feat, int n_agents, int channels, int ny, int nx,
                                  float* canvas, void* stream) {
    HEAL_REQUIRE(cell_map && pillar_feat && canvas && n_agents >= 1 && ny >= 1 && nx >= 1, "pillar_canvas: bad arguments");
    return heal_canvas_from_map(cell_map, pillar_feat, n_agents, channels, ny * nx, canvas, (hipStream_t)stream);
}

static PfnGeom pfn_geom(float vx, float vy, float vz, float xo, float yo, float zo) {
    PfnGeom g;
    g.vx = vx; g.vy = vy; g.vz = vz; g.xo = xo; g.yo = yo; g.zo = zo;
    g.n_agents = 0; g.ny = 0; g.nx = 0;
    return g;
}

// blocks (= rows of the partial-sum outputs) heal_pfn_moments / heal_pfn_backward use for n_voxels pillars
extern "C" int heal_pfn_train_blocks(int n_voxels) {
    const int b = ceil_div(n_voxels < 1 ? 1 : n_voxels, 16);
    return b < 1024 ? b : 1024;
}

extern "C" int heal_pfn_features(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                                 int max_points, const float* weight, const float* bn_scale, const float* bn_shift, float vx,
                                 float vy, float vz, float x_offset, float y_offset, float z_offset, float* pillar_feat,
                                 void* stream) {
    HEAL_REQUIRE(max_points >= 1 && max_points <= 32, "pfn_features: max_points must be in [1,32]");
    HEAL_REQUIRE(n_voxels >= 0 && pillar_feat, "pfn_features: bad arguments");
    if (n_voxels == 0) return 0;
    k_pfn4<<<min(ceil_div(n_voxels, 16), 256 * 16), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(voxels), max_points, reinterpret_cast<const int4*>(coords), num_points, n_voxels, nullptr,
        weight, bn_scale, bn_shift, pfn_geom(vx, vy, vz, x_offset, y_offset, z_offset), pillar_feat, nullptr);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_pfn_moments(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                                int max_points, float vx, float vy, float vz, float x_offset, float y_offset, float z_offset,
                                float* partials, void* stream) {
    HEAL_REQUIRE(max_points >= 1 && max_points <= 32, "pfn_moments: max_points must be in [1,32]");
    HEAL_REQUIRE(n_voxels >= 1 && partials, "pfn_moments: bad arguments");
    k_pfn_moments<<<heal_pfn_train_blocks(n_voxels), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(voxels), max_points, reinterpret_cast<const int4*>(coords), num_points, n_voxels,
        pfn_geom(vx, vy, vz, x_offset, y_offset, z_offset), partials);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_pfn_backward(const float* voxels, const int32_t* coords, const int32_t* num_points, int n_voxels,
                                 int max_points, const float* weight, const float* bn_scale, const float* bn_shift,
                                 const float* mean, const float* rstd, float vx, float vy, float vz, float x_offset,
                                 float y_offset, float z_offset, const float* grad_pillar, float* partials, void* stream) {
    HEAL_REQUIRE(max_points >= 1 && max_points <= 32, "pfn_backward: max_points must be in [1,32]");
    HEAL_REQUIRE(n_voxels >= 1 && grad_pillar && partials && ((uintptr_t)grad_pillar & 15) == 0, "pfn_backward: bad arguments");
    k_pfn_backward<<<heal_pfn_train_blocks(n_voxels), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(voxels), max_points, reinterpret_cast<const int4*>(coords), num_points, n_voxels, weight,
        bn_scale, bn_shift, mean, rstd, pfn_geom(vx, vy, vz, x_offset, y_offset, z_offset), grad_pillar, partials);
    HEAL_LAUNCH_CHECK();
    return 0;
}

