// K4 -- Lift-Splat frustum -> BEV pooling, fused with the depth softmax and the depth (x) feature
// outer product.
//
// Reference arithmetic: opencood/models/heter_encoders.py:125-147 (get_geometry), :161-217
// (voxel_pooling: `.long()` truncation, bounds filter, rank, argsort, cumsum trick, scatter into a
// zero [B,C,Z,Y,X] tensor, Z folded into channels), opencood/utils/camera_utils.py:220-236
// (QuickCumsum.forward) and opencood/models/sub_modules/lss_submodule.py:129-134
// (softmax(depth)[:,None] * feat[:,:,None], a 302 MB/agent tensor the reference materialises).
//
// MI355X formulation -- the 302 MB lifted tensor is never formed.  Two pipelines: the default "splat" (column runs
// as a GEMM on the matrix cores + contiguous fp32 atomics, see k_lss_mark / k_lss_splat_mfma below) and the
// bit-reproducible sorted pipeline (HEAL_LSS_PATH=sorted):
//   k_lss_keys       one thread per camera pixel: softmax over the D depth bins in registers, frustum
//                    geometry in fp32 (same operation order as the reference), cell key per point
//   radix sort       stable sort of (cell key, point index): points of a cell become contiguous, in
//                    point-index order (so every per-cell sum has a fixed, deterministic order)
//   k_lss_segments   segment heads / tails of the sorted keys
//   k_lss_transpose  features to pixel-major [BN, fH*fW, C] so that one point reads C contiguous floats
//   k_lss_reduce     one wave per BEV cell, lanes over channels: sum p_d * f over the cell's points
//                    (sequential fp32, more accurate than the reference's cumsum difference), emit a
//                    compact row and the cell->row map
//   k_canvas         (shared with K2) one streaming pass writes the whole [B, C*nz, ny, nx] output
#include <stdlib.h>
#include <string.h>
#include "prims.h"
#include "../../include/heal_amd.h"

int heal_canvas_from_map(const int* cell_map, const float* rows, int n_agents, int channels, int cells,
                         float* canvas, hipStream_t s);

namespace heal {

struct LssGeom {
    float dx[3], lo[3];  // lo = bx - dx/2
    int nx[3];
    int n_agents, n_cams, D, fH, fW, C;
};

struct CamMats {          // 27 floats per (agent, cam), see heal_amd.h
    float combine[9];     // rots @ inv(intrins)
    float inv_post_rot[9];
    float post_trans[3];
    float trans[3];
    float pad[3];
};

// 64 pixels x 4 depth groups per block: the depth softmax is reduced across the 4 groups through LDS, so
// the (small) pixel count of a camera rig still fills the chip and every logit is read by one thread only.
__global__ __launch_bounds__(256) void k_lss_keys(const float* __restrict__ depth_logit,
                                                 const float* __restrict__ frustum,
                                                 const CamMats* __restrict__ cams, LssGeom g,
                                                 uint32_t invalid_key, uint32_t* __restrict__ keys,
                                                 uint32_t* __restrict__ vals, float* __restrict__ probs) {
    constexpr int DG = 4;
    __shared__ float red[DG][64];
    const int HW = g.fH * g.fW;
    const int px = threadIdx.x & 63, dg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + px;
    const int total = g.n_agents * g.n_cams * HW;
    const bool live = t < total;
    const int bn = live ? t / HW : 0, pix = live ? t - bn * HW : 0;
    const int b = bn / g.n_cams;
    const int dper = (g.D + DG - 1) / DG;
    const int d0 = dg * dper, d1 = min(d0 + dper, g.D);
    const float* lg = depth_logit + (size_t)bn * g.D * HW + pix;
    // softmax over depth (lss_submodule.py:130): max, exp, normalise
    float mx = -INFINITY;
    if (live) for (int d = d0; d < d1; ++d) mx = fmaxf(mx, lg[(size_t)d * HW]);
    red[dg][px] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][px], red[1][px]), fmaxf(red[2][px], red[3][px]));
    __syncthreads();
    float part = 0.f;
    if (live) for (int d = d0; d < d1; ++d) part += expf(lg[(size_t)d * HW] - mx);
    red[dg][px] = part;
    __syncthreads();
    const float den = ((red[0][px] + red[1][px]) + red[2][px]) + red[3][px];
    if (!live) return;
    const CamMats cm = cams[bn];
    const int cells_per_agent = g.nx[0] * g.nx[1] * g.nx[2];
    for (int d = d0; d < d1; ++d) {
        const size_t idx = ((size_t)bn * g.D + d) * HW + pix;
        probs[idx] = expf(lg[(size_t)d * HW] - mx) / den;
        // get_geometry: undo post transform, lift by depth, camera -> ego
        const float* fr = frustum + ((size_t)d * HW + pix) * 3;
        const float p0 = fr[0] - cm.post_trans[0], p1 = fr[1] - cm.post_trans[1], p2 = fr[2] - cm.post_trans[2];
        const float* A = cm.inv_post_rot;
        const float q0 = (A[0] * p0 + A[1] * p1) + A[2] * p2;
        const float q1 = (A[3] * p0 + A[4] * p1) + A[5] * p2;
        const float q2 = (A[6] * p0 + A[7] * p1) + A[8] * p2;
        const float u0 = q0 * q2, u1 = q1 * q2, u2 = q2;
        const float* M = cm.combine;
        const float ex = ((M[0] * u0 + M[1] * u1) + M[2] * u2) + cm.trans[0];
        const float ey = ((M[3] * u0 + M[4] * u1) + M[5] * u2) + cm.trans[1];
        const float ez = ((M[6] * u0 + M[7] * u1) + M[8] * u2) + cm.trans[2];
        // voxel_pooling: ((geom - (bx - dx/2)) / dx).long()  -- truncation toward zero
        const float fx = (ex - g.lo[0]) / g.dx[0];
        const float fy = (ey - g.lo[1]) / g.dx[1];
        const float fz = (ez - g.lo[2]) / g.dx[2];
        uint32_t key = invalid_key;
        // compare in float first so that huge / NaN values never reach the int conversion
        if (fx > -1.f && fx < (float)g.nx[0] && fy > -1.f && fy < (float)g.nx[1] && fz > -1.f &&
            fz < (float)g.nx[2]) {
            const int ix = (int)fx, iy = (int)fy, iz = (int)fz;  // trunc: (-1,0) -> 0 like .long()
            if (ix >= 0 && ix < g.nx[0] && iy >= 0 && iy < g.nx[1] && iz >= 0 && iz < g.nx[2])
                key = (uint32_t)(b * cells_per_agent + (iz * g.nx[1] + iy) * g.nx[0] + ix);
        }
        keys[idx] = key;
        vals[idx] = (uint32_t)idx;
    }
}

__global__ __launch_bounds__(256) void k_lss_segments(const uint32_t* __restrict__ skeys, int n,
                                                     uint32_t invalid_key, int* __restrict__ seg_start,
                                                     int* __restrict__ seg_end) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t k = skeys[j];
    if (k >= invalid_key) return;
    if (j == 0 || skeys[j - 1] != k) seg_start[k] = j;
    if (j == n - 1 || skeys[j + 1] != k) seg_end[k] = j + 1;
}

// [BN, C, HW] -> [BN, HW, C] through a 32x33 LDS tile
__global__ __launch_bounds__(256) void k_lss_transpose(const float* __restrict__ in, int C, int HW,
                                                      float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int bn = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? in[((size_t)bn * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) out[((size_t)bn * HW + p) * C + c] = tile[tx][r];
    }
}

// Segmented reduction over the sorted point list, balanced by construction: a wave owns a TILE of 128
// consecutive sorted points, whatever cells they belong to (a near-range cell can hold thousands of
// points, a far one a handful -- one wave per cell left most of the chip idle behind a few huge cells).
// Runs (points of one cell) that lie entirely inside the tile are finished here; a run that crosses a
// tile boundary leaves a partial row (slot 0: the run began in an earlier tile, slot 1: it begins here and
// continues), and k_lss_combine adds the partials of such a cell in tile order.  Additions happen in a fixed
// order, so the result is deterministic.
constexpr int LSS_TILE = 32;  // points per wave: small tiles = many waves in flight (the reduce is latency-bound)

template <int CPL /*channels per lane*/>
__global__ __launch_bounds__(256) void k_lss_reduce_tiles(const uint32_t* __restrict__ skeys,
                                                         const uint32_t* __restrict__ svals,
                                                         const int* __restrict__ seg_start,
                                                         const int* __restrict__ seg_end,
                                                         const float* __restrict__ probs,
                                                         const float* __restrict__ featT, LssGeom g, int np,
                                                         uint32_t invalid_key,
                                                         float* __restrict__ rows, int* __restrict__ cell_map,
                                                         float* __restrict__ partial /*[ntiles][2][C]*/) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t_begin = tile * LSS_TILE;
    if (t_begin >= np) return;
    const int t_end = min(t_begin + LSS_TILE, np);
    const int l = threadIdx.x & 63;
    const int HW = g.fH * g.fW, DHW = g.D * HW;
    float acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
    uint32_t cur = invalid_key;  // no open run

    auto flush = [&](uint32_t cell) {
        const int s = seg_start[cell], e = seg_end[cell];
        if (s >= t_begin && e <= t_end) {  // run complete inside this tile
            // row = cell: the compact-row counter this replaces was ~9 k same-address atomics (~15 ns each,
            // serialised at the L2) and alone cost more than the whole reduction
            if (l == 0) cell_map[cell] = (int)cell;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = l + 64 * k;
                if (c < g.C) rows[(size_t)cell * g.C + c] = acc[k];
            }
        } else {
            const int slot = (s < t_begin) ? 0 : 1;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = l + 64 * k;
                if (c < g.C) partial[((size_t)tile * 2 + slot) * g.C + c] = acc[k];
            }
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
    };

    for (int base = t_begin; base < t_end; base += 64) {
        const int cnt = min(64, t_end - base);
        uint32_t my_key = invalid_key;
        float my_p = 0.f;
        int my_off = 0;
        if (l < cnt) {
            my_key = skeys[base + l];
            if (my_key < invalid_key) {
                const uint32_t idx = svals[base + l];
                const int bn = idx / DHW;
                const int pix = (idx - bn * DHW) % HW;
                my_p = probs[idx];
                my_off = (bn * HW + pix) * g.C;
            }
        }
        for (int j = 0; j < cnt; j += 4) {
            uint32_t key[4];
            float p[4];
            float v[4][CPL];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = min(j + u, cnt - 1);
                key[u] = (j + u < cnt) ? __shfl(my_key, jj, 64) : invalid_key;
                p[u] = __shfl(my_p, jj, 64);
                const float* f = featT + __shfl(my_off, jj, 64);
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int c = l + 64 * k;
                    v[u][k] = (c < g.C && key[u] < invalid_key) ? f[c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (key[u] >= invalid_key) continue;  // dropped points sort last: nothing valid follows
                if (key[u] != cur) {
                    if (cur != invalid_key) flush(cur);
                    cur = key[u];
                }
#pragma unroll
                for (int k = 0; k < CPL; ++k) acc[k] += p[u] * v[u][k];
            }
        }
    }
    if (cur != invalid_key) flush(cur);
}

// One wave per TILE: a cell whose run starts in this tile and continues past its end is finished here by adding
// the partial rows of the following tiles in tile order (cells finished inside one tile were already written by
// k_lss_reduce_tiles; empty cells keep the -1 of the cell_map memset).
template <int CPL>
__global__ __launch_bounds__(256) void k_lss_combine(const uint32_t* __restrict__ skeys,
                                                    const int* __restrict__ seg_start,
                                                    const int* __restrict__ seg_end,
                                                    const float* __restrict__ partial, int C, int np,
                                                    uint32_t invalid_key,
                                                    float* __restrict__ rows, int* __restrict__ cell_map) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int t_begin = tile * LSS_TILE;
    if (t_begin >= np) return;
    const int t_end = min(t_begin + LSS_TILE, np);
    const uint32_t cell = skeys[t_end - 1];
    if (cell >= invalid_key) return;
    const int s = seg_start[cell], e = seg_end[cell];
    if (e <= t_end || s < t_begin) return;  // ends here, or this tile is not the head of the run
    const int l = threadIdx.x & 63;
    const int t1 = (e - 1) / LSS_TILE;
    float acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = l + 64 * k;
        acc[k] = c < C ? partial[((size_t)tile * 2 + 1) * C + c] : 0.f;
    }
    // a near-camera cell can span hundreds of tiles: 8 partial rows in flight per step, added in tile order
    for (int t = tile + 1; t <= t1; t += 8) {
        float v[8][CPL];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = l + 64 * k;
                v[u][k] = (c < C && t + u <= t1) ? partial[((size_t)(t + u) * 2 + 0) * C + c] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < CPL; ++k) acc[k] += v[u][k];
    }
    if (l == 0) cell_map[cell] = (int)cell;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = l + 64 * k;
        if (c < C) rows[(size_t)cell * C + c] = acc[k];
    }
}

// ---- fused lift + splat (default) --------------------------------------------------------------------------------
// The sorted pipeline above costs ~10 dependent launches for 42 MB of algorithmic traffic.  The frustum has a
// structure the sort ignores: along an image COLUMN (fixed camera, u, depth bin) the lifted points differ only in
// the camera's vertical direction, which the one-cell-high BEV grid collapses -- the fH points of a column fall
// into one or a few cells ("runs").  In the common case a (u, d) column is ONE run, so the column's result is a
// matrix-vector product, and the 16 depth bins of a block share the feature rows:
//     out[d, c] = sum_v P'[d, v] X[v, c],   P'[d, v] = p[d, v] where the point's cell is the column's MAIN cell (the
// cell of its first valid point), 0 elsewhere -- a [16 x fH] x [fH x C] GEMM per block on v_mfma_f32_16x16x4_f32.
// Points of a column that fall in another cell (a pitched camera; none for a level rig) are walked afterwards as runs
// along v.
//
// TWO launches, no memset, no transposition, no separate mark pass:
//   k_lss_scatter  block = (camera, u, 16 depth bins).  Input is the PIXEL-MAJOR head tensor [BN, fH*fW, C + D] the fused
//                  image_head | depth_head convolution writes (heal_conv1x1 out_pixel_major): a pixel's C features and D
//                  depth logits are one contiguous 704-B row, so every load is coalesced.  One wave per image row v:
//                  lanes = depth bins, softmax by wave reductions (once per point), cell key of the block's bins in the
//                  reference's fp32 operation order (lss_cell_key), the column GEMM on the matrix cores, then C
//                  consecutive floats per column added to the cell's row with hardware fp32 atomics
//                  (global_atomic_add_f32: whole cache lines per wave instruction) and the cell tagged flags[cell] = gen.
//   k_lss_canvas   streams the [B, C*nz, ny, nx] output once (K2's canvas writer shape: 4 cells x CG channels per thread,
//                  16-B stores) from the tagged rows and ZEROES every row slice it has read.
// Scratch contract (heal_bev_pool_pm): `rows` is all-zero on entry and all-zero again on exit (self-cleaning), flags hold
// the tag of the call that last touched a cell, state = {generation, tag of the call in flight}: k_lss_scatter tags with
// generation + 1 and publishes the tag, k_lss_canvas adopts it as the new generation, so nothing is ever memset.  Order of the atomic adds across columns is not fixed:
// a cell fed by three or more columns can differ by an ulp from call to call (the reference's unstable `argsort` feeding a
// cumsum difference has the same property); HEAL_LSS_PATH=sorted selects the bit-reproducible pipeline.
constexpr uint32_t LSS_NOKEY = 0xFFFFFFFFu;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int LSS_MT = 16;  // depth bins per block = MFMA M

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// get_geometry + the voxel_pooling index of one lifted point (heter_encoders.py:125-147, :170-186), fp32, the
// operation order of the reference (and of k_lss_keys)
__device__ __forceinline__ uint32_t lss_cell_key(const CamMats& cm, const float* __restrict__ fr, const LssGeom& g,
                                                 int b) {
    const float p0 = fr[0] - cm.post_trans[0], p1 = fr[1] - cm.post_trans[1], p2 = fr[2] - cm.post_trans[2];
    const float* A = cm.inv_post_rot;
    const float q0 = (A[0] * p0 + A[1] * p1) + A[2] * p2;
    const float q1 = (A[3] * p0 + A[4] * p1) + A[5] * p2;
    const float q2 = (A[6] * p0 + A[7] * p1) + A[8] * p2;
    const float u0 = q0 * q2, u1 = q1 * q2, u2 = q2;
    const float* M = cm.combine;
    const float ex = ((M[0] * u0 + M[1] * u1) + M[2] * u2) + cm.trans[0];
    const float ey = ((M[3] * u0 + M[4] * u1) + M[5] * u2) + cm.trans[1];
    const float ez = ((M[6] * u0 + M[7] * u1) + M[8] * u2) + cm.trans[2];
    const float fx = (ex - g.lo[0]) / g.dx[0];
    const float fy = (ey - g.lo[1]) / g.dx[1];
    const float fz = (ez - g.lo[2]) / g.dx[2];
    if (fx > -1.f && fx < (float)g.nx[0] && fy > -1.f && fy < (float)g.nx[1] && fz > -1.f && fz < (float)g.nx[2]) {
        const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        if (ix >= 0 && ix < g.nx[0] && iy >= 0 && iy < g.nx[1] && iz >= 0 && iz < g.nx[2])
            return (uint32_t)(b * (g.nx[0] * g.nx[1] * g.nx[2]) + (iz * g.nx[1] + iy) * g.nx[0] + ix);
    }
    return LSS_NOKEY;
}

// The frustum of create_frustum (heter_encoders.py:110-123) is separable -- frustum[d][v][u] = (xs[u], ys[v], ds[d]) --
// and is read as such (three short axes instead of D*fH*fW strided triples); the host wrapper verifies the property once
// per frustum tensor and takes the sorted pipeline for anything else.
//
// Measured anatomy of the first version of this kernel (m2, 768 blocks, rocprofv3, scripts/k4_dbg.sh): loads 6.6 us,
// softmax + keys 15.3 us, main cells 1.8 us, GEMM + atomics 3.4 us (the atomics themselves: 0.4 us).  The softmax was one
// wave per image row with two ds_bpermute reductions per row and the keys ran on 16 of 64 lanes: a chain of dependent
// long-latency instructions at 3 waves per SIMD.  Now: logits go through an LDS tile (coalesced 192-B rows in, one thread
// per (pixel, quarter of the bins) out), the softmax is a register loop with two block reductions through LDS, and the 768
// keys of a block are 3 independent evaluations per thread.
constexpr int LSS_LGS = 65;   // LDS row stride of the logit tile (floats): lane v reads lgs[v][d] -> bank (v + d) % 32

__global__ __launch_bounds__(256) void k_lss_scatter(const float* __restrict__ head /*[BN, HW, CT]*/, int CT,
                                                    const float* __restrict__ frustum,
                                                    const CamMats* __restrict__ cams, LssGeom g, int n_dt,
                                                    float* __restrict__ rows, int* __restrict__ flags,
                                                    int* __restrict__ state, int dbg) {
    __shared__ float red[4][64];
    __shared__ float pk_p[LSS_MT][64];       // p[dl][v] (unmasked; the leftover walk reads it)
    __shared__ uint32_t pk_key[LSS_MT][64];  // key[dl][v]; NOKEY for v >= fH and bins >= D
    __shared__ float pT[64][LSS_MT];         // P'[v][dl]: A operand, v-major so that a fragment read is conflict-free
    __shared__ uint32_t mk[LSS_MT];          // main cell of column dl
    __shared__ int has_left[LSS_MT];
    extern __shared__ float4 xs4[];          // X[fH4][C + 16], then the logit tile lgs[fH][LSS_LGS]
    float* xs = reinterpret_cast<float*>(xs4);

    const int HW = g.fH * g.fW;
    const int u = blockIdx.x / n_dt, dt = blockIdx.x % n_dt, bn = blockIdx.y;
    const int part = threadIdx.x >> 6, l = threadIdx.x & 63, v = l;
    const int LD = g.C + 16;                 // LD % 64 == 16: the 4 k-rows of a B fragment hit disjoint bank groups
    const int fH4 = (g.fH + 3) & ~3;
    float* lgs = xs + (size_t)fH4 * LD;
    const int gen = state[0] + 1;            // tag of this call: >= 1, a zero-filled flag array matches nothing
    const float* __restrict__ hcol = head + ((size_t)bn * HW + u) * CT;  // pixel (v, u) = hcol + v * fW * CT
    const bool live = v < g.fH;

    // stage the column's feature rows (512-B contiguous runs; rows fH .. fH4-1 are zero: they meet P' = 0 in the k loop) and
    // its logit rows (D contiguous floats per pixel).  ALL global loads of a thread are issued before the first LDS store:
    // a load -> store loop with a run-time trip count serialises one HBM round trip per iteration (6.6 us of the first
    // version of this kernel).
    {
        const int c4n = g.C / 4, ld4 = LD / 4, nX = fH4 * c4n;
        const int d4n = g.D / 4, nL = g.fH * d4n;   // D % 4 == 0 (host)
        float4 xr[8], lr[4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = threadIdx.x + 256 * k, r = i / c4n, c4 = i - r * c4n;
            xr[k] = (i < nX && r < g.fH) ? *reinterpret_cast<const float4*>(hcol + (size_t)r * g.fW * CT + c4 * 4)
                                         : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, r = i / d4n, d4 = i - r * d4n;
            lr[k] = i < nL ? *reinterpret_cast<const float4*>(hcol + (size_t)r * g.fW * CT + g.C + d4 * 4)
                           : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = threadIdx.x + 256 * k, r = i / c4n, c4 = i - r * c4n;
            if (i < nX) xs4[r * ld4 + c4] = xr[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k, r = i / d4n, d4 = i - r * d4n;
            if (i < nL) {
                float* dst = lgs + r * LSS_LGS + d4 * 4;
                dst[0] = lr[k].x; dst[1] = lr[k].y; dst[2] = lr[k].z; dst[3] = lr[k].w;
            }
        }
        for (int i = threadIdx.x + 2048; i < nX; i += 256) {   // C > 160: the rest of the feature rows
            const int r = i / c4n, c4 = i - r * c4n;
            xs4[r * ld4 + c4] = r < g.fH ? *reinterpret_cast<const float4*>(hcol + (size_t)r * g.fW * CT + c4 * 4)
                                         : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // cell keys of the block's 16 bins: thread (v, part) takes dl = part, part + 4, part + 8, part + 12 (independent chains)
    {
        const CamMats cm = cams[bn];
        const float fr_x = frustum[(size_t)u * 3 + 0];
        const float fr_y = live ? frustum[(size_t)v * g.fW * 3 + 1] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dl = part + 4 * j, d = dt * LSS_MT + dl;
            uint32_t key = LSS_NOKEY;
            if (live && d < g.D) {
                const float fr[3] = {fr_x, fr_y, frustum[(size_t)d * HW * 3 + 2]};
                key = lss_cell_key(cm, fr, g, bn / g.n_cams);
            }
            pk_key[dl][v] = key;
        }
    }
    __syncthreads();
    if (dbg & 16) return;
    // softmax over depth (lss_submodule.py:130): thread (v, part) owns the bins [part * dper, part * dper + dper)
    const int dper = (g.D + 3) / 4;          // <= 16
    float e[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = part * dper + i;
        e[i] = (live && i < dper && d < g.D) ? lgs[v * LSS_LGS + d] : -INFINITY;
        mx = fmaxf(mx, e[i]);
    }
    red[part][v] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][v], red[1][v]), fmaxf(red[2][v], red[3][v]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        e[i] = e[i] != -INFINITY ? expf(e[i] - mx) : 0.f;
        sum += e[i];
    }
    red[part][v] = sum;
    __syncthreads();
    const float den = ((red[0][v] + red[1][v]) + red[2][v]) + red[3][v];
    // probabilities of the block's bins that this thread owns
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int dl = part * dper + i - dt * LSS_MT;
        if (i < dper && dl >= 0 && dl < LSS_MT) pk_p[dl][v] = (live && pk_key[dl][v] != LSS_NOKEY) ? e[i] / den : 0.f;
    }
    // bins of the tile beyond D (D not a multiple of 16) belong to nobody above: p = 0
    for (int dl = part; dl < LSS_MT; dl += 4)
        if (dt * LSS_MT + dl >= g.D) pk_p[dl][v] = 0.f;
    if (dbg & 32) return;
    __syncthreads();
    // main cell of each column: the cell of its first valid point; tag it
    for (int dl = part; dl < LSS_MT; dl += 4) {
        const uint32_t key = pk_key[dl][l];
        const unsigned long long valid = __ballot(key != LSS_NOKEY);
        const uint32_t m = valid ? __shfl(key, __builtin_ctzll(valid), 64) : LSS_NOKEY;
        const unsigned long long left = __ballot(key != LSS_NOKEY && key != m);
        if (l == 0) {
            mk[dl] = m;
            has_left[dl] = left != 0ull;
            if (m != LSS_NOKEY) flags[m] = gen;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dl = part + 4 * j;
        const uint32_t key = pk_key[dl][l];
        pT[l][dl] = (key != LSS_NOKEY && key == mk[dl]) ? pk_p[dl][l] : 0.f;
    }
    __syncthreads();
    if (dbg & 64) return;

    // GEMM: D[dl, c] = sum_v P'[v][dl] X[v][c]; wave w owns the channel tiles w, w+4, ...
    const int lk = l >> 4, ln = l & 15;
    const int ksteps = fH4 / 4;  // <= 16
    float afr[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) afr[ks] = ks < ksteps ? pT[ks * 4 + lk][ln] : 0.f;
    const int n_tiles = g.C / 16;
    for (int nt = part; nt < n_tiles; nt += 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* xb = xs + lk * LD + nt * 16 + ln;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            if (ks < ksteps) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], xb[ks * 4 * LD], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t cell = mk[lk * 4 + r];
            if (cell != LSS_NOKEY) unsafeAtomicAdd(rows + (size_t)cell * g.C + nt * 16 + ln, acc[r]);
        }
    }

    // leftovers: points of a column outside its main cell, walked as runs along v
    for (int dl = part; dl < LSS_MT; dl += 4) {
        if (!has_left[dl]) continue;
        const uint32_t skip = mk[dl];
        for (int c0 = 0; c0 < g.C; c0 += 64) {
            const int c = c0 + l;
            float acc = 0.f;
            uint32_t cur = LSS_NOKEY;
            for (int vv = 0; vv <= g.fH; ++vv) {
                uint32_t key = vv < g.fH ? pk_key[dl][vv] : LSS_NOKEY;
                if (key == skip) key = LSS_NOKEY;
                if (key != cur) {
                    if (cur != LSS_NOKEY) {
                        if (c < g.C) unsafeAtomicAdd(rows + (size_t)cur * g.C + c, acc);
                        if (l == 0 && c0 == 0) flags[cur] = gen;
                    }
                    acc = 0.f;
                    cur = key;
                }
                if (cur != LSS_NOKEY && c < g.C) acc += pk_p[dl][vv] * xs[vv * LD + c];
            }
        }
    }

    // Publish the tag for k_lss_canvas in state[1] -- a word nobody READS in this kernel (every block derives the tag from
    // state[0]); k_lss_canvas reads state[1] and writes it back to state[0], which nobody reads there.  The kernel boundary
    // orders the two: no tickets, no returning atomics (a last-block ticket cost 7-10 us in either kernel: the returning
    // atomic waits behind the block's outstanding stores / atomics).
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) state[1] = gen;
}

// canvas[b][c][cell] = flags[b][cell] == generation ? rows[b*cells + cell][c] : 0.  A thread owns 4 consecutive x cells x 4
// channels (one float4 row read per tagged cell, a 4x4 register transpose, four 16-B stores); the four channel slices of a
// 16-channel group sit in ADJACENT LANES, so a wave's row reads -- and the zero stores that clean the rows behind the reads
// -- cover whole 64-B lines (16-B partial-line stores from separate blocks cost 4 us of the first version, the last-block
// ticket that used to live in this kernel 10 us: rocprofv3, scripts/k4_dbg.sh).
__global__ __launch_bounds__(256) void k_lss_canvas(const int4* __restrict__ flags4, float* __restrict__ rows, int cells4,
                                                   int channels, float4* __restrict__ canvas4,
                                                   int* __restrict__ state, int dbg) {
    const int gen = state[1];
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) state[0] = gen;
    const int s = threadIdx.x & 3;
    const int t = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int b = blockIdx.z, c0 = blockIdx.y * 16 + s * 4;
    if (t >= cells4) return;
    const int4 f = flags4[(size_t)b * cells4 + t];
    float4* out = canvas4 + ((size_t)b * channels + c0) * cells4 + t;
    const bool h0 = f.x == gen, h1 = f.y == gen, h2 = f.z == gen, h3 = f.w == gen;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(h0 | h1 | h2 | h3)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[(size_t)c * cells4] = z;
        return;
    }
    float4* r = reinterpret_cast<float4*>(rows + ((size_t)b * cells4 * 4 + (size_t)t * 4) * channels + c0);
    const size_t rs = channels / 4;  // float4 stride between consecutive cells' rows
    const float4 a0 = h0 ? r[0] : z, a1 = h1 ? r[rs] : z, a2 = h2 ? r[2 * rs] : z, a3 = h3 ? r[3 * rs] : z;
    out[0] = make_float4(a0.x, a1.x, a2.x, a3.x);
    out[(size_t)cells4] = make_float4(a0.y, a1.y, a2.y, a3.y);
    out[(size_t)2 * cells4] = make_float4(a0.z, a1.z, a2.z, a3.z);
    out[(size_t)3 * cells4] = make_float4(a0.w, a1.w, a2.w, a3.w);
    if (!(dbg & 2)) {
        if (h0) r[0] = z;
        if (h1) r[rs] = z;
        if (h2) r[2 * rs] = z;
        if (h3) r[3 * rs] = z;
    }
}

// One thread per camera: the 3x3 algebra of get_geometry (closed-form adjugate inverses, fp32) in ONE launch instead
// of ~90 tiny elementwise kernels per camera modality.
__device__ __forceinline__ void inv3x3(const float* m, float* o) {
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const float A = e * i - f * h, B = c * h - b * i, C = b * f - c * e;
    const float D = f * g - d * i, E = a * i - c * g, F = c * d - a * f;
    const float G = d * h - e * g, H = b * g - a * h, I = a * e - b * d;
    const float det = a * A + b * D + c * G;
    o[0] = A / det; o[1] = B / det; o[2] = C / det;
    o[3] = D / det; o[4] = E / det; o[5] = F / det;
    o[6] = G / det; o[7] = H / det; o[8] = I / det;
}

__global__ __launch_bounds__(64) void k_camera_matrices(const float* __restrict__ rots, const float* __restrict__ trans,
                                                       const float* __restrict__ intrins,
                                                       const float* __restrict__ post_rots,
                                                       const float* __restrict__ post_trans, int n,
                                                       float* __restrict__ out /*[n,27]*/) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= n) return;
    float ii[9], R[9], K[9], P[9], ip[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { R[k] = rots[t * 9 + k]; K[k] = intrins[t * 9 + k]; P[k] = post_rots[t * 9 + k]; }
    inv3x3(K, ii);
    inv3x3(P, ip);
    float* o = out + (size_t)t * 27;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)  // combine = rots @ inv(intrins), summed left to right like a matmul row
            o[r * 3 + c] = (R[r * 3 + 0] * ii[0 * 3 + c] + R[r * 3 + 1] * ii[1 * 3 + c]) + R[r * 3 + 2] * ii[2 * 3 + c];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[9 + k] = ip[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[18 + k] = post_trans[t * 3 + k]; o[21 + k] = trans[t * 3 + k]; o[24 + k] = 0.f; }
}

struct LssWs {
    uint32_t *keys[2], *vals[2];
    float *probs, *featT, *rows, *partial;
    int *seg_start, *seg_end, *cell_map, *scratch;
};

static bool carve(Arena& a, int n_agents, int n_cams, int D, int HW, int C, int cells_total, LssWs& w) {
    const size_t np = (size_t)n_agents * n_cams * D * HW;
    for (int k = 0; k < 2; ++k) { w.keys[k] = a.take<uint32_t>(np); w.vals[k] = a.take<uint32_t>(np); }
    w.probs = a.take<float>(np);
    w.featT = a.take<float>((size_t)n_agents * n_cams * HW * C);
    w.rows = a.take<float>(((size_t)cells_total + 1) * C);  // row of a cell = its index (only non-empty rows are touched)
    w.partial = a.take<float>((size_t)(np / LSS_TILE + 2) * 2 * C);
    // seg_start | seg_end contiguous: one memset clears both
    w.seg_start = a.take<int>(cells_total);
    w.seg_end = a.take<int>(cells_total);
    w.cell_map = a.take<int>(cells_total);
    w.scratch = a.take<int>(sort_scratch_words((int64_t)np));
    return a.ok();
}

}  // namespace heal

using namespace heal;

extern "C" int heal_camera_matrices(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                                    const float* post_trans, int n_cameras, float* cam_mats, void* stream) {
    HEAL_REQUIRE(n_cameras >= 1, "camera_matrices: no cameras");
    HEAL_REQUIRE(rots && trans && intrins && post_rots && post_trans && cam_mats, "camera_matrices: null pointer");
    k_camera_matrices<<<ceil_div(n_cameras, 64), 64, 0, (hipStream_t)stream>>>(rots, trans, intrins, post_rots, post_trans,
                                                                            n_cameras, cam_mats);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_bev_pool_workspace(int n_agents, int n_cams, int D, int fH, int fW, int channels,
                                          int nx, int ny, int nz) {
    Arena a(nullptr, 0);
    LssWs w;
    carve(a, n_agents, n_cams, D, fH * fW, channels, n_agents * nx * ny * nz, w);
    return a.off + 256;
}

extern "C" int heal_bev_pool(const float* depth_logit, const float* feat, const float* frustum,
                             const float* cam_mats, int n_agents, int n_cams, int D, int fH, int fW,
                             int channels, const float* dx_host, const float* bx_host,
                             const int32_t* nx_host, float* out, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && n_cams >= 1 && D >= 1 && fH >= 1 && fW >= 1, "bev_pool: bad shape");
    HEAL_REQUIRE(channels >= 16 && channels <= 256 && channels % 16 == 0,
                 "bev_pool: channels must be a multiple of 16 in [16,256] (got %d)", channels);
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "bev_pool: workspace must be 256-B aligned");
    LssGeom g;
    for (int k = 0; k < 3; ++k) {
        g.dx[k] = dx_host[k];
        g.lo[k] = bx_host[k] - dx_host[k] / 2.f;  // (self.bx - self.dx/2.) in fp32
        g.nx[k] = nx_host[k];
        HEAL_REQUIRE(g.nx[k] >= 1, "bev_pool: empty grid");
    }
    g.n_agents = n_agents; g.n_cams = n_cams; g.D = D; g.fH = fH; g.fW = fW; g.C = channels;
    const int HW = fH * fW;
    const int cells_per_agent = g.nx[0] * g.nx[1] * g.nx[2];
    HEAL_REQUIRE((g.nx[0] * g.nx[1]) % 4 == 0, "bev_pool: nx*ny must be a multiple of 4");
    const int64_t cells_total64 = (int64_t)cells_per_agent * n_agents;
    const int64_t np64 = (int64_t)n_agents * n_cams * D * HW;
    HEAL_REQUIRE(cells_total64 < (1ll << 30) && np64 < (1ll << 31), "bev_pool: problem too large");
    const int cells_total = (int)cells_total64, np = (int)np64;
    Arena a(ws, ws_bytes);
    LssWs w;
    HEAL_REQUIRE(carve(a, n_agents, n_cams, D, HW, channels, cells_total, w),
                 "bev_pool: workspace too small (%zu < %zu)", ws_bytes, a.off);

    HEAL_HIP(hipMemsetAsync(w.cell_map, 0xFF, (size_t)cells_total * sizeof(int), s));  // -1 = empty cell
    const uint32_t invalid_key = (uint32_t)cells_total;
    k_lss_keys<<<ceil_div(n_agents * n_cams * HW, 64), 256, 0, s>>>(
        depth_logit, frustum, reinterpret_cast<const CamMats*>(cam_mats), g,
                                                                     invalid_key, w.keys[0], w.vals[0], w.probs);
    k_lss_transpose<<<dim3(ceil_div(HW, 32), ceil_div(channels, 32), n_agents * n_cams), 256, 0, s>>>(
        feat, channels, HW, w.featT);
    int key_bits = 1;
    while ((1u << key_bits) <= invalid_key) ++key_bits;
    int res = 0;
    if (radix_sort_pairs(w.keys, w.vals, np, key_bits, &res, w.scratch, s)) return 1;
    k_lss_segments<<<ceil_div(np, 256), 256, 0, s>>>(w.keys[res], np, invalid_key, w.seg_start, w.seg_end);
    const int tblocks = ceil_div(ceil_div(np, LSS_TILE), 4);
#define HEAL_LSS_REDUCE(CPL)                                                                                   \
    k_lss_reduce_tiles<CPL><<<tblocks, 256, 0, s>>>(w.keys[res], w.vals[res], w.seg_start, w.seg_end, w.probs,  \
                                                    w.featT, g, np, invalid_key, w.rows,        \
                                                    w.cell_map, w.partial);                                    \
    k_lss_combine<CPL><<<tblocks, 256, 0, s>>>(w.keys[res], w.seg_start, w.seg_end, w.partial, channels, np,    \
                                               invalid_key, w.rows, w.cell_map)
    if (channels <= 64) { HEAL_LSS_REDUCE(1); }
    else if (channels <= 128) { HEAL_LSS_REDUCE(2); }
    else { HEAL_LSS_REDUCE(4); }
#undef HEAL_LSS_REDUCE
    HEAL_LAUNCH_CHECK();
    // out [B, C*nz, ny, nx] viewed as B*nz maps of [C, ny*nx]
    return heal_canvas_from_map(w.cell_map, w.rows, n_agents * g.nx[2], channels, g.nx[0] * g.nx[1], out, s);
}


// ---- pixel-major entry point (the production path) ---------------------------------------------------------------------
namespace heal {
struct LssPmWs { float* rows; int* flags; int* state; };
static bool carve_pm(Arena& a, int channels, int cells_total, LssPmWs& w) {
    w.state = a.take<int>(64);                                   // {generation, tag of the call in flight}
    w.flags = a.take<int>(cells_total);
    w.rows = a.take<float>(((size_t)cells_total + 1) * channels);
    return a.ok();
}
}  // namespace heal

extern "C" size_t heal_bev_pool_pm_workspace(int n_agents, int channels, int nx, int ny, int nz) {
    Arena a(nullptr, 0);
    LssPmWs w;
    carve_pm(a, channels, n_agents * nx * ny * nz, w);
    return a.off + 256;
}

extern "C" int heal_bev_pool_pm(const float* head, int head_stride, const float* frustum, const float* cam_mats,
                                int n_agents, int n_cams, int D, int fH, int fW, int channels, const float* dx_host,
                                const float* bx_host, const int32_t* nx_host, float* out, void* ws, size_t ws_bytes,
                                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && n_cams >= 1 && D >= 1 && fH >= 1 && fW >= 1, "bev_pool_pm: bad shape");
    HEAL_REQUIRE(channels >= 16 && channels <= 256 && channels % 16 == 0,
                 "bev_pool_pm: channels must be a multiple of 16 in [16,256] (got %d)", channels);
    HEAL_REQUIRE(fH <= 64 && D <= 64, "bev_pool_pm: fH and D must be <= 64 (got %d, %d): use heal_bev_pool", fH, D);
    HEAL_REQUIRE(head_stride >= channels + D && head_stride % 4 == 0 && ((uintptr_t)head & 15) == 0,
                 "bev_pool_pm: head rows must hold C + D floats, 16-B aligned (stride %d)", head_stride);
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "bev_pool_pm: workspace must be 256-B aligned");
    HEAL_REQUIRE(D % 4 == 0, "bev_pool_pm: D must be a multiple of 4 (got %d)", D);
    const size_t lds = ((size_t)((fH + 3) & ~3) * (channels + 16) + (size_t)fH * LSS_LGS) * sizeof(float);
    HEAL_REQUIRE(lds <= 64 * 1024 - 14 * 1024, "bev_pool_pm: feature column of %zu B does not fit the LDS budget: use heal_bev_pool", lds);
    LssGeom g;
    for (int k = 0; k < 3; ++k) {
        g.dx[k] = dx_host[k];
        g.lo[k] = bx_host[k] - dx_host[k] / 2.f;  // (self.bx - self.dx/2.) in fp32
        g.nx[k] = nx_host[k];
        HEAL_REQUIRE(g.nx[k] >= 1, "bev_pool_pm: empty grid");
    }
    g.n_agents = n_agents; g.n_cams = n_cams; g.D = D; g.fH = fH; g.fW = fW; g.C = channels;
    HEAL_REQUIRE((g.nx[0] * g.nx[1]) % 4 == 0, "bev_pool_pm: nx*ny must be a multiple of 4");
    const int64_t cells_total64 = (int64_t)g.nx[0] * g.nx[1] * g.nx[2] * n_agents;
    HEAL_REQUIRE(cells_total64 < (1ll << 30), "bev_pool_pm: problem too large");
    const int cells_total = (int)cells_total64;
    Arena a(ws, ws_bytes);
    LssPmWs w;
    HEAL_REQUIRE(carve_pm(a, channels, cells_total, w), "bev_pool_pm: workspace too small (%zu < %zu)", ws_bytes, a.off);
    const int n_dt = ceil_div(D, LSS_MT);
    const char* dbg_env = getenv("HEAL_K4_DBG");   // timing experiments only (bits skip parts of the work: results invalid)
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    k_lss_scatter<<<dim3(fW * n_dt, n_agents * n_cams), 256, lds, s>>>(head, head_stride, frustum,
                                                                      reinterpret_cast<const CamMats*>(cam_mats), g, n_dt,
                                                                      w.rows, w.flags, w.state, dbg);
    const int cells4 = g.nx[0] * g.nx[1] / 4;
    const dim3 grid(ceil_div(cells4, 64), channels / 16, n_agents * g.nx[2]);
    k_lss_canvas<<<grid, 256, 0, s>>>(reinterpret_cast<const int4*>(w.flags), w.rows, cells4, channels,
                                      reinterpret_cast<float4*>(out), w.state, dbg);
    HEAL_LAUNCH_CHECK();
    return 0;
}

namespace heal {
// ---- backward of the lift + splat (training) ----------------------------------------------------------------
// Forward: out[cell(cam, d, v, u)][c] += p[cam, d, v, u] * f[cam, c, v, u],  p = softmax_d(logit[cam, :, v, u]).
// Given G = dL/d out (re-laid cell-major [cells][C] by the caller so that a cell's gradient is one contiguous row):
//     dL/df[c]      = sum_d p_d G[cell_d][c]
//     dL/dlogit_d   = p_d (s_d - sum_d' p_d' s_d'),   s_d = sum_c f[c] G[cell_d][c]
// One wave per image pixel: lanes hold the D logits (softmax by wave reductions) and C/64 feature channels each; the cell of
// every depth bin comes from lss_cell_key -- the SAME fp32 arithmetic as the forward kernels, so forward and backward agree on
// which cell a point belongs to; the D bins are walked with the row gather of bin d+1 independent of the reduction of bin d.
__global__ __launch_bounds__(256) void k_lss_backward(const float* __restrict__ gcells, const float* __restrict__ depth_logit,
                                                     const float* __restrict__ feat, const float* __restrict__ frustum,
                                                     const CamMats* __restrict__ cams, LssGeom g,
                                                     float* __restrict__ glogit, float* __restrict__ gfeat) {
    constexpr int CPL = 4;                       // channels per lane (C <= 256)
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int HWf = g.fH * g.fW;
    const long long pix = (long long)blockIdx.x * 4 + wave;
    if (pix >= (long long)g.n_agents * g.n_cams * HWf) return;   // wave-uniform
    const int bn = (int)(pix / HWf), vu = (int)(pix - (long long)bn * HWf);
    const int v = vu / g.fW, u = vu - v * g.fW, b = bn / g.n_cams;
    const float lg = l < g.D ? depth_logit[((size_t)bn * g.D + l) * HWf + vu] : -INFINITY;
    const float mx = wave_max(lg);
    const float e = l < g.D ? expf(lg - mx) : 0.f;
    const float p = e / wave_sum(e);
    uint32_t key = LSS_NOKEY;
    if (l < g.D) key = lss_cell_key(cams[bn], frustum + ((size_t)(l * g.fH + v) * g.fW + u) * 3, g, b);
    float f[CPL], gf[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = l + 64 * k;
        f[k] = c < g.C ? feat[((size_t)bn * g.C + c) * HWf + vu] : 0.f;
        gf[k] = 0.f;
    }
    float s_mine = 0.f;
    for (int d = 0; d < g.D; ++d) {
        const uint32_t kd = __shfl(key, d, 64);
        if (kd == LSS_NOKEY) continue;               // wave-uniform
        const float pd = __shfl(p, d, 64);
        const float* __restrict__ row = gcells + (size_t)kd * g.C;
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = l + 64 * k;
            const float gv = c < g.C ? row[c] : 0.f;
            gf[k] = fmaf(pd, gv, gf[k]);
            part = fmaf(f[k], gv, part);
        }
        const float s = wave_sum(part);
        if (l == d) s_mine = s;
    }
    const float dot = wave_sum(p * s_mine);
    if (l < g.D) glogit[((size_t)bn * g.D + l) * HWf + vu] = p * (s_mine - dot);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = l + 64 * k;
        if (c < g.C) gfeat[((size_t)bn * g.C + c) * HWf + vu] = gf[k];
    }
}

}  // namespace heal

extern "C" int heal_bev_pool_backward(const float* grad_cells, const float* depth_logit, const float* feat,
                                      const float* frustum, const float* cam_mats, int n_agents, int n_cams, int D, int fH,
                                      int fW, int channels, const float* dx_host, const float* bx_host,
                                      const int32_t* nx_host, float* grad_logit, float* grad_feat, void* stream) {
    using namespace heal;
    HEAL_REQUIRE(n_agents >= 1 && n_cams >= 1 && D >= 1 && D <= 64 && fH >= 1 && fW >= 1, "bev_pool_backward: bad shape (D <= 64)");
    HEAL_REQUIRE(channels >= 1 && channels <= 256, "bev_pool_backward: channels must be in [1, 256] (got %d)", channels);
    HEAL_REQUIRE(grad_cells && depth_logit && feat && frustum && cam_mats && grad_logit && grad_feat, "bev_pool_backward: null pointer");
    LssGeom g;
    for (int k = 0; k < 3; ++k) {
        g.dx[k] = dx_host[k];
        g.lo[k] = bx_host[k] - dx_host[k] / 2.f;
        g.nx[k] = nx_host[k];
        HEAL_REQUIRE(g.nx[k] >= 1, "bev_pool_backward: empty grid");
    }
    g.n_agents = n_agents; g.n_cams = n_cams; g.D = D; g.fH = fH; g.fW = fW; g.C = channels;
    const long long pixels = (long long)n_agents * n_cams * fH * fW;
    HEAL_REQUIRE(pixels < (1ll << 31) - 8, "bev_pool_backward: problem too large");
    k_lss_backward<<<(unsigned)((pixels + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        grad_cells, depth_logit, feat, frustum, reinterpret_cast<const CamMats*>(cam_mats), g, grad_logit, grad_feat);
    HEAL_LAUNCH_CHECK();
    return 0;
}
