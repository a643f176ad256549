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
#ifdef HEAL_BUILD_EXPERIMENTAL
#include "../../include/heal_amd_experimental.h"
#endif

int heal_canvas_from_map(const int* cell_map, const float* rows, int n_agents, int channels, int cells,
                         float* canvas, hipStream_t s);

namespace heal {

struct LssGeom {
    float dx[3], lo[3];  // lo = bx - dx/2
    float rdx[3];        // fl(1 / dx) (lss_grid_coord)
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
// matrix-vector product, and the depth bins of an image column share the feature rows:
//     out[d, c] = sum_v P'[d, v] X[v, c],   P'[d, v] = p[d, v] where the point's cell is the column's MAIN cell (the
// cell of its first valid point), 0 elsewhere -- a [D x fH] x [fH x C] GEMM per image column on v_mfma_f32_16x16x4_f32.
// Points of a column that fall in another cell (a pitched camera; none for a level rig) are walked afterwards as runs
// along v.
//
// Round 4: ONE launch, and the dense canvas is no longer part of K4.
//   k_lss_scatter  block = (camera, image column u) with 256 * ceil(D / 16) threads: every depth bin of the column in one
//                  block, so the column's feature rows are staged ONCE and the depth softmax is evaluated ONCE (round 3: one
//                  block per 16 bins -> three copies of both; staging + keys were 9.2 of the kernel's 19.6 us, softmax 4.1:
//                  scripts/k4_dbg.sh).  Input is the PIXEL-MAJOR head tensor [BN, fH*fW, C + D] of the fused
//                  image_head | depth_head convolution (heal_conv1x1 out_pixel_major): a pixel's C features and D depth
//                  logits are one contiguous row.  Work is split by wave role, not by barrier-separated phases:
//                    * the first ceil(fH / 16) waves hold one pixel per lane QUAD: each lane loads its quarter of the
//                      pixel's logits straight from global memory (16-B loads), softmax = register loop + two quad
//                      butterflies (no LDS tile, no block reduction, no barrier) -> p[d][v] in LDS;
//                    * every thread evaluates D*fH / blockDim cell keys (lss_cell_key, the reference's fp32 operation
//                      order) from frustum values it requested BEFORE the feature loads, while those are in flight;
//                    * the feature rows go global -> registers -> LDS once.
//                  ONE barrier; main cell of every depth bin by ballot (its wave also writes the bin's A-operand row P');
//                  ONE barrier; the [16 ceil(D/16) x fH] x [fH x C] GEMM on the matrix cores, C consecutive floats per column
//                  added to the main cell's row with hardware fp32 atomics (whole cache lines per wave instruction), the cell
//                  tagged flags[cell] = generation.
// Output = the SPARSE PIXEL-MAJOR BEV map of the workspace: rows[cell][C] (cell = ((b*nz + z)*ny + y)*nx + x) + flags[cell].
// A level rig touches <= n_cams * fW * D cells (12 288 of 65 536 at BASELINE size): 6.3 MB instead of the 33.5 MB dense
// [C, ny, nx] canvas, 81 % of which is zeros.  Consumers:
//   heal_bev_stem_block  the first BasicBlock convolutions of ResNetBEVBackbone (3x3 stride 2 + the 1x1 stride-2 downsample)
//                        read the rows through the flags -- the canvas is never materialised (k_bev_stem below);
//   heal_bev_pool_emit   any other consumer: one streaming pass writes the dense [B, C*nz, ny, nx] tensor (k_lss_canvas).
// Scratch contract: TWO halves (rows, flags), selected by the parity of the call's generation.  k_lss_scatter of generation g
// adds into half g & 1 (all-zero rows by induction); its CONSUMER zeroes, as side work of its own launch, the rows of the OTHER
// half that generation g - 1 tagged (lss_clean_other_half: that half's consumer finished launches ago, and the stores hide
// under the consumer's own MFMA / streaming work -- in the scatter they cost 2 of 16 us), so nothing is ever memset and no
// kernel cleans rows that a neighbouring block may still be reading.  state = {generation of the last consumed call, tag of the call in
// flight}: the scatter derives its tag from state[0] and publishes it in state[1]; the consumer (exactly one per scatter)
// reads state[1] and writes it back to state[0] -- each word is only written in the kernel that does not read it, the kernel
// boundary orders them: no tickets, no returning atomics.  Order of the atomic adds across columns is not fixed: a cell fed by
// three or more columns can differ by an ulp from call to call (the reference's unstable `argsort` feeding a cumsum
// difference has the same property); HEAL_LSS_PATH=sorted selects the bit-reproducible pipeline.
constexpr uint32_t LSS_NOKEY = 0xFFFFFFFFu;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int LSS_MT = 16;   // depth bins per MFMA m-tile

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
constexpr int LSS_PKS = 66;  // row stride (floats) of the p / key tables: conflict-free for the softmax quads' writes and the A-fragment reads

// (geom - (bx - dx/2)) / dx of voxel_pooling (heter_encoders.py:170) is only ever truncated and compared with integers.  The IEEE
// quotient q = fl(a / b) costs ~13 instructions, three per lifted point; t = fl(a * fl(1 / b)) lies within 1.5 * 2^-23 |q| of it, so
// trunc(t) = trunc(q) and every comparison of t with an integer equals that of q UNLESS an integer lies that close to t -- only
// then (about one point in 10^4) is the division evaluated, behind a WAVE-UNIFORM branch (a per-lane branch is if-converted and
// the division runs for everybody).  The value used is NOT q, but it truncates and compares like q.
__device__ __forceinline__ bool lss_near_integer(float t) { return fabsf(t - rintf(t)) <= 4e-7f * fmaxf(fabsf(t), 1.f); }

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
    const float ax = ex - g.lo[0], ay = ey - g.lo[1], az = ez - g.lo[2];
    float fx = ax * g.rdx[0], fy = ay * g.rdx[1], fz = az * g.rdx[2];
    const bool nx_ = lss_near_integer(fx), ny_ = lss_near_integer(fy), nz_ = lss_near_integer(fz);
    if (__builtin_amdgcn_ballot_w64(nx_ | ny_ | nz_) != 0ull) {
        if (nx_) fx = ax / g.dx[0];
        if (ny_) fy = ay / g.dx[1];
        if (nz_) fz = az / g.dx[2];
    }
    if (fx > -1.f && fx < (float)g.nx[0] && fy > -1.f && fy < (float)g.nx[1] && fz > -1.f && fz < (float)g.nx[2]) {
        const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        if (ix >= 0 && ix < g.nx[0] && iy >= 0 && iy < g.nx[1] && iz >= 0 && iz < g.nx[2])
            return (uint32_t)(b * (g.nx[0] * g.nx[1] * g.nx[2]) + (iz * g.nx[1] + iy) * g.nx[0] + ix);
    }
    return LSS_NOKEY;
}

struct LssPmWs { float* rows[2]; int* flags[2]; int* state; };

// LDS carve of k_lss_scatter (floats); shared by the kernel and the host's size computation
struct LssLds {
    int LD, fH4, MTOT, xs, pk_p, pk_key, total;
    __host__ __device__ LssLds(int fH, int C, int ndt) {
        LD = C + 16;                         // LD % 64 == 16: the 4 k-rows of a B fragment hit disjoint bank groups
        fH4 = (fH + 3) & ~3;
        MTOT = LSS_MT * ndt;
        xs = 0;
        pk_p = xs + fH4 * LD;
        pk_key = pk_p + MTOT * LSS_PKS;
        total = pk_key + MTOT * LSS_PKS;
    }
};

// The frustum of create_frustum (heter_encoders.py:110-123) is separable -- frustum[d][v][u] = (xs[u], ys[v], ds[d]) --
// and is read as such (three short axes instead of D*fH*fW strided triples); the host wrapper verifies the property once
// per frustum tensor and takes the sorted pipeline for anything else.
// The block's work: column u of camera bn (channel slice zs of nz); `first` = the block that publishes the tag (one per problem).
template <int NDT>
__device__ __forceinline__ void lss_scatter_body(const float* __restrict__ head /*[BN, HW, CT]*/, int CT,
                                                 const float* __restrict__ frustum, const CamMats* __restrict__ cams,
                                                 const LssGeom& g, const LssPmWs& ws, int cells_total, int dbg, int u, int bn, int zs,
                                                 int nz, bool first) {
    constexpr int W = 4 * NDT, NTHR = 256 * NDT, MTOT = LSS_MT * NDT;
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    // zs: channel slice [c_lo, c_lo + Cb) of the column (nz slices): with two slices two half-size blocks share a
    // CU and run out of phase (loads of one under the atomics of the other) at the price of evaluating keys and softmax twice
    const int Cb = g.C / nz, c_lo = zs * Cb;
    const LssLds L(g.fH, Cb, NDT);
    float* xs = smem + L.xs;
    float* pk_p = smem + L.pk_p;
    uint32_t* pk_key = reinterpret_cast<uint32_t*>(smem + L.pk_key);

    const int HW = g.fH * g.fW;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int LD = L.LD, fH4 = L.fH4;
    const int gen = ws.state[0] + 1;         // tag of this call: >= 1, a zero-filled flag array matches nothing
    // HEAL_K4_DBG & 128: shader-clock stamps of the first block's first (softmax) and last (keys only) wave -> state[16..] (scripts/k4_stamps.py)
    const bool stamp_on = (dbg & 128) && first && (threadIdx.x == 0 || threadIdx.x == blockDim.x - 64);
    int* stamp_out = ws.state + 16 + (threadIdx.x == 0 ? 0 : 16);
    const unsigned long long stamp0 = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
#define LSS_STAMP(i) do { if (stamp_on) { __builtin_amdgcn_s_waitcnt(0); stamp_out[i] = (int)(__builtin_amdgcn_s_memtime() - stamp0); } } while (0)
    float* __restrict__ rows = ws.rows[gen & 1];
    int* __restrict__ flags = ws.flags[gen & 1];
    const float* __restrict__ hcol = head + ((size_t)bn * HW + u) * CT;  // pixel (v, u) = hcol + v * fW * CT

    // -- requests, oldest first (vmcnt retires in order): frustum axes of this thread's keys, the softmax waves' logits, the
    //    feature rows.  Every load is UNCONDITIONAL on a clamped address and masked afterwards: a predicated load sits in its own
    //    basic block, and the waitcnt pass then drains the whole queue (vmcnt(0)) in front of the first use of ANY of them.
    const CamMats cm = cams[bn];             // scalar loads: requested first, their latency hides under the vector loads
    const int kb = bn / g.n_cams;
    const int n_keys = g.D * g.fH;
    float kfy[4], kfz[4];
    int kd[4], kv[4];
    const float fr_x = frustum[(size_t)u * 3 + 0];
    {   // key i = tid + NTHR j -> (d, v) = (i / fH, i % fH): one division per thread, then steps of (NTHR / fH, NTHR % fH)
        const int qs = NTHR / g.fH, rs = NTHR - qs * g.fH;    // wave-uniform: scalar unit
        int dcur = tid / g.fH, vcur = tid - dcur * g.fH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {            // D * fH <= 16 NDT * 64 = 4 * NTHR
            kd[j] = min(dcur, g.D - 1);          // clamped: the loads below are unconditional (keys past n_keys are not stored)
            kv[j] = vcur;
            kfy[j] = frustum[(size_t)kv[j] * g.fW * 3 + 1];
            kfz[j] = frustum[(size_t)kd[j] * HW * 3 + 2];
            dcur += qs;
            vcur += rs;
            if (vcur >= g.fH) { vcur -= g.fH; ++dcur; }
        }
    }
    const int n_sw = (g.fH + 15) >> 4;       // softmax waves: one pixel per lane quad
    const int sv = w * 16 + (l >> 2), sq = l & 3, d4n = g.D >> 2;   // D % 4 == 0 (host)
    const bool s_on = w < n_sw && sv < g.fH;
    float4 lg[4];
    if (w < n_sw) {
        const float* lrow = hcol + (size_t)min(sv, g.fH - 1) * g.fW * CT + g.C;
#pragma unroll
        for (int k = 0; k < 4; ++k) lg[k] = *reinterpret_cast<const float4*>(lrow + min(sq + 4 * k, d4n - 1) * 4);
    }
    const int c4n = Cb / 4, ld4 = LD / 4, nX = fH4 * c4n;
    float4 xr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = min(tid + NTHR * k, nX - 1), r = i / c4n, c4 = i - r * c4n;
        xr[k] = *reinterpret_cast<const float4*>(hcol + (size_t)min(r, g.fH - 1) * g.fW * CT + c_lo + c4 * 4);
    }

    LSS_STAMP(0);
    // -- cell keys (VALU work under the feature loads)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (tid + NTHR * j < n_keys) {
                const float fr[3] = {fr_x, kfy[j], kfz[j]};
                pk_key[kd[j] * LSS_PKS + kv[j]] = lss_cell_key(cm, fr, g, kb);
            }
        }
    }
    LSS_STAMP(1);
    // -- softmax over depth (lss_submodule.py:130) in the lane quads of the first waves
    if (w < n_sw && !(dbg & 32)) {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(s_on && sq + 4 * k < d4n)) lg[k] = float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            mx = fmaxf(fmaxf(mx, fmaxf(lg[k].x, lg[k].y)), fmaxf(lg[k].z, lg[k].w));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float e[16];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x4[4] = {lg[k].x, lg[k].y, lg[k].z, lg[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e[4 * k + c] = x4[c] != -INFINITY ? __expf(x4[c] - mx) : 0.f;   // v_exp_f32: 2 ulp, far inside the 1e-3 of the BEV features
                sum += e[4 * k + c];
            }
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float inv = 1.f / sum;             // one division per pixel; p = e * inv is within an ulp of e / sum
        if (s_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i4 = sq + 4 * k;
                if (i4 < d4n) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) pk_p[(i4 * 4 + c) * LSS_PKS + sv] = e[4 * k + c] * inv;
                }
            }
        }
    }
    LSS_STAMP(2);
    // -- feature rows to LDS (rows fH .. fH4-1 are zero: they meet P' = 0 in the k loop)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + NTHR * k, r = i / c4n, c4 = i - r * c4n;
        if (i < nX) smem4[r * ld4 + c4] = r < g.fH ? xr[k] : float4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid + 4 * NTHR; i < nX; i += NTHR) {   // fH4 * C / 4 > 4 blockDim: the rest of the feature rows
        const int r = i / c4n, c4 = i - r * c4n;
        smem4[r * ld4 + c4] = r < g.fH ? *reinterpret_cast<const float4*>(hcol + (size_t)r * g.fW * CT + c_lo + c4 * 4)
                                       : float4{0.f, 0.f, 0.f, 0.f};
    }
    LSS_STAMP(3);
    __syncthreads();
    LSS_STAMP(4);
    if (dbg & 16) return;

    // -- per wave (m-tile w / 4 = 16 depth bins, channel tiles w % 4, w % 4 + 4, ...): main cells, A operand and GEMM with no
    //    further barrier.  Lane (lk, ln) is lane (k = lk, row ln) of the 16x16x4 A fragment: it owns depth bin d = 16 mt + ln and
    //    the image rows v = 4 ks + lk.  Main cell of a bin = the cell of the column's first valid point: the first valid key
    //    among the lane's own rows, then the lowest v of the four lk lanes (two xor shuffles).  The four waves of an m-tile
    //    each derive it (12 LDS reads per lane, the same count the A fragments took from a shared P' table) -- that table, the
    //    main-cell pass over it and the barrier between them are gone.
    const int lk = l >> 4, ln = l & 15;
    const int ksteps = fH4 / 4;  // <= 16
    const int mt = w >> 2, nq = w & 3;
    const int d = mt * LSS_MT + ln;
    uint32_t kk[16];
    float pp[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int v = 4 * ks + lk;
        const bool on = ks < ksteps && d < g.D && v < g.fH;
        kk[ks] = on ? pk_key[d * LSS_PKS + v] : LSS_NOKEY;
        pp[ks] = on ? pk_p[d * LSS_PKS + v] : 0.f;
    }
    int fv = 1 << 20;                        // lowest image row with a valid key among this lane's rows, and its key
    uint32_t m = LSS_NOKEY;
#pragma unroll
    for (int ks = 15; ks >= 0; --ks)
        if (kk[ks] != LSS_NOKEY) { fv = 4 * ks + lk; m = kk[ks]; }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const int fo = __shfl_xor(fv, o, 64);
        const uint32_t mo = __shfl_xor(m, o, 64);
        if (fo < fv) { fv = fo; m = mo; }
    }
    bool left = false;
    float afr[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        left |= kk[ks] != LSS_NOKEY && kk[ks] != m;
        afr[ks] = (kk[ks] != LSS_NOKEY && kk[ks] == m) ? pp[ks] : 0.f;
    }
    LSS_STAMP(5);
    if (nq == 0 && lk == 0 && m != LSS_NOKEY) flags[m] = gen;
    if (dbg & 64) return;
    uint32_t cell_r[4];                      // D[row = 4 lk + r][col = ln]: the main cell of bin 16 mt + 4 lk + r lives in lane 4 lk + r
#pragma unroll
    for (int r = 0; r < 4; ++r) cell_r[r] = __shfl(m, lk * 4 + r, 64);
    const int n_tiles = Cb / 16;
    for (int nt = nq; nt < n_tiles; nt += 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* xb = xs + lk * LD + nt * 16 + ln;
        // all B fragments of the tile first (ONE wait), then the MFMAs in groups of 4 k-steps; k-steps past fH4 / 4 meet A = 0
        // and read a clamped (valid) feature row.  (A guard per MFMA puts every LDS read behind its own wait.)
        float bq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) bq[q] = xb[min(q, ksteps - 1) * 4 * LD];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (g4 * 4 < ksteps) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g4 * 4 + q], bq[g4 * 4 + q], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t cell = cell_r[r];
            if (dbg & 2) continue;
            if (dbg & 4) { if (cell != LSS_NOKEY) rows[(size_t)cell * g.C + c_lo + nt * 16 + ln] = acc[r]; continue; }
            if (cell != LSS_NOKEY) unsafeAtomicAdd(rows + (size_t)cell * g.C + c_lo + nt * 16 + ln, acc[r]);
        }
    }

    LSS_STAMP(6);
    // -- leftovers: points of a column outside its main cell (a pitched camera; none for a level rig), walked as runs along v by
    //    the first wave of the m-tile, 64 channels per pass
    if (nq == 0) {
        unsigned long long todo = __ballot(left) ;
        todo = (todo | (todo >> 16) | (todo >> 32) | (todo >> 48)) & 0xFFFFull;     // bins (ln) with a leftover in any lk lane
        while (todo) {
            const int bl = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int dd = mt * LSS_MT + bl;
            const uint32_t skip = __shfl(m, bl, 64);
            for (int c0 = 0; c0 < Cb; c0 += 64) {
                const int c = c0 + l;
                float acc = 0.f;
                uint32_t cur = LSS_NOKEY;
                for (int vv = 0; vv <= g.fH; ++vv) {
                    uint32_t key = vv < g.fH ? pk_key[dd * LSS_PKS + vv] : LSS_NOKEY;
                    if (key == skip) key = LSS_NOKEY;
                    if (key != cur) {
                        if (cur != LSS_NOKEY) {
                            if (c < Cb) unsafeAtomicAdd(rows + (size_t)cur * g.C + c_lo + c, acc);
                            if (l == 0 && c0 == 0) flags[cur] = gen;
                        }
                        acc = 0.f;
                        cur = key;
                    }
                    if (cur != LSS_NOKEY && c < Cb) acc += pk_p[dd * LSS_PKS + vv] * xs[vv * LD + c];
                }
            }
        }
    }

    LSS_STAMP(7);
    // Publish the tag for the consumer in state[1] -- a word nobody READS in this kernel (every block derives the tag from
    // state[0]); the consumer reads state[1] and writes it back to state[0], which nobody reads there.
    if (first && tid == 0) ws.state[1] = gen;
#undef LSS_STAMP
}

template <int NDT>
__global__ __launch_bounds__(256 * NDT) void k_lss_scatter(const float* __restrict__ head /*[BN, HW, CT]*/, int CT,
                                                          const float* __restrict__ frustum,
                                                          const CamMats* __restrict__ cams, LssGeom g, LssPmWs ws,
                                                          int cells_total, int dbg) {
    lss_scatter_body<NDT>(head, CT, frustum, cams, g, ws, cells_total, dbg, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z,
                          blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
}

// Round 6: the camera agents of ALL camera modalities of a scene in ONE launch (m2 and m4 of the heterogeneous scene have different image
// sizes, heads, frustums and workspaces but the same kernel): a block finds its problem by its index range.  One launch ramp / tail
// instead of one per modality, and the blocks of the second problem fill the CUs the first leaves idle (256 + 224 columns: one round of
// two resident blocks per CU) -- the per-agent fraction of VERDICT r5 item 3a.
#ifdef HEAL_BUILD_EXPERIMENTAL
constexpr int LSS_MAX_PROBLEMS = 4;
struct LssProblem {
    const float* head; const float* frustum; const CamMats* cams;
    LssGeom g; LssPmWs ws;
    int CT, cells_total, blocks;              // blocks = columns x cameras of this problem
};

// TWO problems per launch, each with its own inlined copy of the body: the problem descriptors stay kernel ARGUMENTS read with scalar loads.
// (A first version took an array of descriptors and selected one by the block's index: the compiler copied the selected struct into vector
// registers and scratch -- 123 registers + 160 B of scratch against 85 + 0 -- and the launch took 46.7 us against 2 x 14 for two launches.)
template <int NDT>
__global__ __launch_bounds__(256 * NDT) void k_lss_scatter_pair(LssProblem p0, LssProblem p1, int dbg) {
    const int b = blockIdx.x;
    if (b < p0.blocks) {
        lss_scatter_body<NDT>(p0.head, p0.CT, p0.frustum, p0.cams, p0.g, p0.ws, p0.cells_total, dbg, b % p0.g.fW, b / p0.g.fW, 0, 1, b == 0);
    } else {
        const int c = b - p0.blocks;
        lss_scatter_body<NDT>(p1.head, p1.CT, p1.frustum, p1.cams, p1.g, p1.ws, p1.cells_total, dbg, c % p1.g.fW, c / p1.g.fW, 0, 1, c == 0);
    }
}

#endif

// Zero the rows that generation gen - 1 tagged in the half the call in flight does NOT use; `part` of `nparts` equal cell ranges
// per block.  Called by the consumer kernels (every block, all threads; blockDim a multiple of 64).
__device__ __forceinline__ void lss_clean_other_half(const LssPmWs& ws, int gen, int cells_total, int channels, int part,
                                                     int nparts) {
    const int cpb = (cells_total + nparts - 1) / nparts;
    const int cbeg = part * cpb, cend = min(cbeg + cpb, cells_total);
    const int* __restrict__ flags_o = ws.flags[(gen & 1) ^ 1];
    float4* __restrict__ rows_o4 = reinterpret_cast<float4*>(ws.rows[(gen & 1) ^ 1]);
    const int c4n = channels / 4, l = threadIdx.x & 63;
    const float4 z = float4{0.f, 0.f, 0.f, 0.f};
    for (int base = cbeg + (int)(threadIdx.x & ~63u); base < cend; base += blockDim.x) {
        const int f = flags_o[min(base + l, cend - 1)];
        unsigned long long m = __ballot(gen > 1 && base + l < cend && f == gen - 1);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            for (int c4 = l; c4 < c4n; c4 += 64) rows_o4[(size_t)(base + j) * c4n + c4] = z;
        }
    }
}

// Dense emit: canvas[b][c][cell] = flags[cell] == generation ? rows[cell][c] : 0 for consumers that want the reference's
// [B, C*nz, ny, nx] tensor.  A thread owns 4 consecutive x cells x 4 channels (one float4 row read per tagged cell, a 4x4
// register transpose, four 16-B stores); the four channel slices of a 16-channel group sit in ADJACENT LANES, so a wave's row
// reads cover whole 64-B lines.
__global__ __launch_bounds__(256) void k_lss_canvas(LssPmWs ws, int cells4, int channels, float4* __restrict__ canvas4,
                                                   int cells_total) {
    const int gen = ws.state[1];
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) ws.state[0] = gen;
    lss_clean_other_half(ws, gen, cells_total, channels, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                         gridDim.x * gridDim.y * gridDim.z);
    const int4* __restrict__ flags4 = reinterpret_cast<const int4*>(ws.flags[gen & 1]);
    const float* __restrict__ rows = ws.rows[gen & 1];
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
    const float4* r = reinterpret_cast<const float4*>(rows + ((size_t)b * cells4 * 4 + (size_t)t * 4) * channels + c0);
    const size_t rs = channels / 4;  // float4 stride between consecutive cells' rows
    float4 a0 = z, a1 = z, a2 = z, a3 = z;
    if (h0) a0 = r[0];
    if (h1) a1 = r[rs];
    if (h2) a2 = r[2 * rs];
    if (h3) a3 = r[3 * rs];
    out[0] = make_float4(a0.x, a1.x, a2.x, a3.x);
    out[(size_t)cells4] = make_float4(a0.y, a1.y, a2.y, a3.y);
    out[(size_t)2 * cells4] = make_float4(a0.z, a1.z, a2.z, a3.z);
    out[(size_t)3 * cells4] = make_float4(a0.w, a1.w, a2.w, a3.w);
}

// ---- first block of the camera BEV backbone on the sparse pixel-major map ------------------------------------------------
// ResNetBEVBackbone of a camera modality opens with BasicBlock(C -> 64, stride 2) (base_bev_backbone_resnet.py:88-109,
// resblock.py:18-64): conv1 = 3x3 stride 2 pad 1 (+BN+ReLU) and downsample = 1x1 stride 2 (+BN) both read the pooled map.
// One launch computes both from the rows / flags of k_lss_scatter, so the [C, ny, nx] canvas never exists:
//   implicit GEMM, M = 64 output pixels of one output row, N = 64 conv1 channels (+ 64 downsample channels at the centre tap),
//   K = 9 taps x C, on the 32x32x2 fp32 MFMA core of heal_linear: both operands staged [row][k] (k contiguous, row stride 36).
//   A row (an input pixel at a tap) is one contiguous 128-B run of the cell's row -- or zero when the cell is not tagged with
//   the generation in flight (the flags of a thread's two pixels x 9 taps are read once, up front); taps none of the block's
//   64 pixels has are skipped (the map is 81 % empty: outside the cameras' 50 m radius whole tiles are).
//   wave (wm, wn): pixels [32 wm, +32) x channels [32 wn, +32) of conv1 AND of the downsample: one 32x32 accumulator each.
//   K chunks of BK = 64 channels (32 when C % 64 != 0), two accumulators per output so that back-to-back MFMAs are independent.
//   Weights pre-laid by the host: main [9][C/BK][64][BK] (tap, chunk, cout, k), downsample [C/BK][64][BK].
//   Epilogue through LDS: bias (BN folded), ReLU on the conv1 half, NCHW 16-B stores.
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int BK>   // channels per K chunk: 64 (C % 64 == 0) or 32
__global__ __launch_bounds__(256) void k_bev_stem(LssPmWs ws, int C, int nx, int ny, int cells_total,
                                                 const float* __restrict__ wmain, const float* __restrict__ bmain,
                                                 const float* __restrict__ wds, const float* __restrict__ bds,
                                                 float* __restrict__ out_main, float* __restrict__ out_id, int Ho, int Wo,
                                                 int tiles_x, int dbg) {
    constexpr int RS = BK + 4;               // LDS row stride: 16-B aligned rows, the b128 reads of 16 lanes cover all banks
    constexpr int NH = BK / 32;              // 128-B pieces of a row chunk per thread
    constexpr int KH = BK / 2;               // k-steps per chunk (one 32x32x2 MFMA each: lane half h multiplies k = KH h + s)
    __shared__ __attribute__((aligned(16))) float s_all[4 * 64 * RS];
    __shared__ int s_tapmask;
    float (*sA)[64 * RS] = reinterpret_cast<float (*)[64 * RS]>(s_all);
    float (*sB)[64 * RS] = reinterpret_cast<float (*)[64 * RS]>(s_all + 2 * 64 * RS);
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, li = l & 31, h = l >> 5;
    const int gen = ws.state[1];
    if (blockIdx.x == 0 && tid == 0) ws.state[0] = gen;
    const float* __restrict__ rows = ws.rows[gen & 1];
    const int* __restrict__ flags = ws.flags[gen & 1];
    int bx_ = blockIdx.x;
    const int xt = bx_ % tiles_x; bx_ /= tiles_x;
    const int oy = bx_ % Ho, b = bx_ / Ho;
    const int ox0 = xt * 64;
    if (tid == 0) s_tapmask = 0;

    // staging role: pixel rows r0 and r0 + 32 of the A tile, 16-B column c4 of each 128-B piece; weight rows likewise.  The
    // flags of the thread's two pixels x 9 taps are read ONCE (18 unconditional loads on clamped cells, masked afterwards: see
    // k_lss_scatter).
    const int r0 = tid >> 3, c4 = tid & 7;
    int pm0 = 0, pm1 = 0;
    const int cell_max = (b + 1) * ny * nx - 1, cell_min = b * ny * nx;
    const int cl0 = (b * ny + 2 * oy - 1) * nx + 2 * (ox0 + r0) - 1, cl1 = cl0 + 64;   // cell of tap (0, 0) of the two pixels
    {
        int f0[9], f1[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f0[t] = flags[min(max(cl0 + (t / 3) * nx + t % 3, cell_min), cell_max)];
            f1[t] = flags[min(max(cl1 + (t / 3) * nx + t % 3, cell_min), cell_max)];
        }
        // side work: the rows the PREVIOUS generation tagged in the other half go back to zero (stores that drain under the MFMAs)
        lss_clean_other_half(ws, gen, cells_total, C, blockIdx.x, gridDim.x);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = 2 * oy + t / 3 - 1, ix0 = 2 * (ox0 + r0) + t % 3 - 1, ix1 = ix0 + 64;
            const bool iny = iy >= 0 && iy < ny;
            pm0 |= (int)(iny && ox0 + r0 < Wo && ix0 >= 0 && ix0 < nx && f0[t] == gen) << t;
            pm1 |= (int)(iny && ox0 + r0 + 32 < Wo && ix1 >= 0 && ix1 < nx && f1[t] == gen) << t;
        }
    }
    __syncthreads();
    {
        int m = pm0 | pm1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
        if (l == 0 && m) atomicOr(&s_tapmask, m);
    }
    __syncthreads();
    // virtual tap 9 = the centre tap once more, against the DOWNSAMPLE weights: every iteration then stages one 64 x BK A tile
    // and one 64 x BK weight tile, and only the accumulators it feeds differ (a wave-uniform choice)
    int tapmask = s_tapmask;
    tapmask |= ((tapmask >> 4) & 1) << 9;
    const int nch = C / BK;
    const int n_it = (dbg & 2048) ? 0 : __popc(tapmask) * nch;

    f32x16 acc_m[2], acc_d[2];               // two accumulators per output: consecutive MFMAs never wait for each other
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_m[0][r] = 0.f; acc_m[1][r] = 0.f; acc_d[0][r] = 0.f; acc_d[1][r] = 0.f; }

    // iteration -> (tap, chunk): taps in ascending order over the set bits of tapmask
    int it_tap = 0, it_ch = 0, rem = tapmask;
    auto first_tap = [&]() { it_tap = rem ? __builtin_ctz(rem) : 0; rem &= rem - 1; it_ch = 0; };
    auto next_it = [&]() { if (++it_ch == nch) first_tap(); };
    // staging registers as NAMED float4 values (arrays of float4 written in one branch and read in another end up in scratch);
    // the second 128-B piece (x1) exists for BK = 64 only
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a00 = z4, a01 = z4, a10 = z4, a11 = z4, b0 = z4, b1 = z4, b2 = z4, b3 = z4;
    auto load = [&](int t, int j) {
        const int tp = t == 9 ? 4 : t;        // pixel tap
        const int dy = tp / 3, dx = tp - dy * 3;
        const float* wsrc = t == 9 ? wds + ((size_t)j * 64) * BK : wmain + ((size_t)(t * nch + j) * 64) * BK;
        const int c0 = ((pm0 >> tp) & 1) ? cl0 + dy * nx + dx : cell_min;
        const int c1 = ((pm1 >> tp) & 1) ? cl1 + dy * nx + dx : cell_min;
        const float* p0 = rows + (size_t)c0 * C + j * BK + c4 * 4;
        const float* p1 = rows + (size_t)c1 * C + j * BK + c4 * 4;
        if (!(dbg & 512)) {
            a00 = *reinterpret_cast<const float4*>(p0);
            a10 = *reinterpret_cast<const float4*>(p1);
        }
        if (!(dbg & 1024)) {
            b0 = *reinterpret_cast<const float4*>(wsrc + tid * 4);
            b1 = *reinterpret_cast<const float4*>(wsrc + (tid + 256) * 4);
        }
        if constexpr (NH == 2) {
            if (!(dbg & 512)) {
                a01 = *reinterpret_cast<const float4*>(p0 + 32);
                a11 = *reinterpret_cast<const float4*>(p1 + 32);
            }
            if (!(dbg & 1024)) {
                b2 = *reinterpret_cast<const float4*>(wsrc + (tid + 512) * 4);
                b3 = *reinterpret_cast<const float4*>(wsrc + (tid + 768) * 4);
            }
        }
    };
    auto put_a = [&](float* dst, const float4& v, bool on) {   // per component: float4 selects are lowered through scratch
        *reinterpret_cast<float4*>(dst) = make_float4(on ? v.x : 0.f, on ? v.y : 0.f, on ? v.z : 0.f, on ? v.w : 0.f);
    };
    auto put_b = [&](float* sb, int q, const float4& v) {      // weight tile [64][BK] contiguous: float4 index tid + 256 q
        const int i4 = tid + 256 * q, row = i4 / (BK / 4), col4 = i4 % (BK / 4);
        *reinterpret_cast<float4*>(sb + row * RS + col4 * 4) = v;
    };
    auto store = [&](int buf, int t) {        // the mask is applied HERE: a select next to the load would wait for it at once
        const int tp = t == 9 ? 4 : t;
        const bool on0 = (pm0 >> tp) & 1, on1 = (pm1 >> tp) & 1;
        put_a(&sA[buf][r0 * RS + c4 * 4], a00, on0);
        put_a(&sA[buf][(r0 + 32) * RS + c4 * 4], a10, on1);
        put_b(sB[buf], 0, b0);
        put_b(sB[buf], 1, b1);
        if constexpr (NH == 2) {
            put_a(&sA[buf][r0 * RS + 32 + c4 * 4], a01, on0);
            put_a(&sA[buf][(r0 + 32) * RS + 32 + c4 * 4], a11, on1);
            put_b(sB[buf], 2, b2);
            put_b(sB[buf], 3, b3);
        }
    };
    if (n_it > 0) {
        first_tap();
        load(it_tap, it_ch);
        store(0, it_tap);
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1, t_cur = it_tap;
        next_it();                                        // (it_tap, it_ch) now name iteration it + 1
        if (it + 1 < n_it) load(it_tap, it_ch);           // in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        float af[KH], bf[KH];
        {
            const float* p = &sA[buf][(wm * 32 + li) * RS + KH * h];
            const float* q = &sB[buf][(wn * 32 + li) * RS + KH * h];
#pragma unroll
            for (int k = 0; k < KH / 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * k);
                af[4 * k] = v.x; af[4 * k + 1] = v.y; af[4 * k + 2] = v.z; af[4 * k + 3] = v.w;
                const float4 u = *reinterpret_cast<const float4*>(q + 4 * k);
                bf[4 * k] = u.x; bf[4 * k + 1] = u.y; bf[4 * k + 2] = u.z; bf[4 * k + 3] = u.w;
            }
        }
        if (dbg & 256) {
        } else if (t_cur == 9) {
#pragma unroll
            for (int s = 0; s < KH; ++s) acc_d[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc_d[s & 1], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < KH; ++s) acc_m[s & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc_m[s & 1], 0, 0, 0);
        }
        if (it + 1 < n_it) store(buf ^ 1, it_tap);
        __syncthreads();
    }

    // epilogue: accumulators (C/D layout of 32x32: column = lane & 31 = channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = pixel)
    // -> sC[channel][pixel] (stride 65: conflict-free writes) -> 16-B NCHW stores; the conv1 half first, then the downsample half
    constexpr int CS = 65;
    float* sC = s_all;
    static_assert(sizeof(s_all) >= (size_t)64 * CS * 4, "epilogue tile must fit the operand rings");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sC[(wn * 32 + li) * CS + px] = half ? acc_d[0][r] + acc_d[1][r] : acc_m[0][r] + acc_m[1][r];
        }
        __syncthreads();
        const float* __restrict__ bias = half ? bds : bmain;
        float* __restrict__ outp = half ? out_id : out_main;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i, co = idx >> 4, p4 = idx & 15, px = p4 * 4;
            if (ox0 + px >= Wo) continue;         // Wo % 4 == 0 (host)
            const float* sp = &sC[co * CS + px];
            const float bv = bias ? bias[co] : 0.f;
            float4 v = make_float4(sp[0] + bv, sp[1] + bv, sp[2] + bv, sp[3] + bv);
            if (!half) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(outp + (((size_t)b * 64 + co) * Ho + oy) * Wo + ox0 + px) = v;
        }
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
        g.rdx[k] = 1.f / dx_host[k];
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

    HEAL_FILL(w.cell_map, 0xFF, (size_t)cells_total * sizeof(int), s);  // -1 = empty cell
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


// ---- pixel-major entry points (the production path) ---------------------------------------------------------------------
namespace heal {
static bool carve_pm(Arena& a, int channels, int cells_total, LssPmWs& w) {
    w.state = a.take<int>(64);                                   // {generation consumed, tag of the call in flight}
    for (int h = 0; h < 2; ++h) w.flags[h] = a.take<int>(cells_total);
    for (int h = 0; h < 2; ++h) w.rows[h] = a.take<float>((size_t)cells_total * channels);
    return a.ok();
}

static int pm_geometry(int n_agents, int channels, const float* dx_host, const float* bx_host, const int32_t* nx_host,
                       LssGeom& g, int& cells_total) {
    HEAL_REQUIRE(channels >= 16 && channels <= 256 && channels % 16 == 0,
                 "bev_pool_pm: channels must be a multiple of 16 in [16,256] (got %d)", channels);
    for (int k = 0; k < 3; ++k) {
        g.dx[k] = dx_host ? dx_host[k] : 1.f;
        g.rdx[k] = 1.f / g.dx[k];
        g.lo[k] = dx_host ? bx_host[k] - dx_host[k] / 2.f : 0.f;  // (self.bx - self.dx/2.) in fp32
        g.nx[k] = nx_host[k];
        HEAL_REQUIRE(g.nx[k] >= 1, "bev_pool_pm: empty grid");
    }
    HEAL_REQUIRE((g.nx[0] * g.nx[1]) % 4 == 0, "bev_pool_pm: nx*ny must be a multiple of 4");
    const int64_t cells_total64 = (int64_t)g.nx[0] * g.nx[1] * g.nx[2] * n_agents;
    HEAL_REQUIRE(n_agents >= 1 && cells_total64 < (1ll << 30), "bev_pool_pm: problem too large");
    cells_total = (int)cells_total64;
    return 0;
}

template <int NDT>
static int launch_scatter(const float* head, int head_stride, const float* frustum, const float* cam_mats, const LssGeom& g,
                          const LssPmWs& w, int cells_total, size_t lds, int csplit, int dbg, hipStream_t s) {
    static bool attr_set = false;   // > 64 KB of dynamic LDS needs the attribute once per kernel (gfx950: 160 KB per CU)
    if (!attr_set) {
        HEAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lss_scatter<NDT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    HEAL_LAUNCH_EV(k_lss_scatter<NDT>, dim3(g.fW, g.n_agents * g.n_cams, csplit), dim3(256 * NDT), lds, s, head, head_stride, frustum,
                   reinterpret_cast<const CamMats*>(cam_mats), g, w, cells_total, dbg);
    return 0;
}
}  // namespace heal

extern "C" size_t heal_bev_pool_pm_workspace(int n_agents, int channels, int nx, int ny, int nz) {
    Arena a(nullptr, 0);
    LssPmWs w;
    carve_pm(a, channels, n_agents * nx * ny * nz, w);
    return a.off + 256;
}

extern "C" int heal_bev_pool_scatter(const float* head, int head_stride, const float* frustum, const float* cam_mats,
                                     int n_agents, int n_cams, int D, int fH, int fW, int channels, const float* dx_host,
                                     const float* bx_host, const int32_t* nx_host, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_agents >= 1 && n_cams >= 1 && D >= 1 && fH >= 1 && fW >= 1, "bev_pool_pm: bad shape");
    HEAL_REQUIRE(fH <= 64 && D <= 64, "bev_pool_pm: fH and D must be <= 64 (got %d, %d): use heal_bev_pool", fH, D);
    HEAL_REQUIRE(head_stride >= channels + D && head_stride % 4 == 0 && ((uintptr_t)head & 15) == 0,
                 "bev_pool_pm: head rows must hold C + D floats, 16-B aligned (stride %d)", head_stride);
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "bev_pool_pm: workspace must be 256-B aligned");
    HEAL_REQUIRE(D % 4 == 0, "bev_pool_pm: D must be a multiple of 4 (got %d)", D);
    HEAL_REQUIRE(n_agents * n_cams <= 65535, "bev_pool_pm: too many cameras for one launch");
    LssGeom g;
    int cells_total = 0;
    if (pm_geometry(n_agents, channels, dx_host, bx_host, nx_host, g, cells_total)) return 1;
    g.n_agents = n_agents; g.n_cams = n_cams; g.D = D; g.fH = fH; g.fW = fW; g.C = channels;
    const int n_dt = ceil_div(D, LSS_MT);
    // channel slices per column (grid.z).  Measured NEGATIVE at BASELINE size (HEAL_K4_CSPLIT=2: 14.9 vs 13.3 us for m2, 13.8 vs
    // 12.1 for m4): the duplicated keys + softmax cost more than the two half-size blocks per CU gain by running out of phase.
    int csplit = 1;
    if (const char* e = getenv("HEAL_K4_CSPLIT")) csplit = atoi(e) == 2 && channels % 32 == 0 ? 2 : 1;
    const size_t lds = (size_t)LssLds(fH, channels / csplit, n_dt).total * sizeof(float);
    HEAL_REQUIRE(lds <= 150 * 1024, "bev_pool_pm: feature column of %zu B does not fit the LDS: use heal_bev_pool", lds);
    Arena a(ws, ws_bytes);
    LssPmWs w;
    HEAL_REQUIRE(carve_pm(a, channels, cells_total, w), "bev_pool_pm: workspace too small (%zu < %zu)", ws_bytes, a.off);
    const int dbg = HEAL_DEBUG_ENV("HEAL_K4_DBG");   // timing experiments only (bits skip parts of the work: results invalid)
    int rc = 1;
    switch (n_dt) {
        case 1: rc = launch_scatter<1>(head, head_stride, frustum, cam_mats, g, w, cells_total, lds, csplit, dbg, s); break;
        case 2: rc = launch_scatter<2>(head, head_stride, frustum, cam_mats, g, w, cells_total, lds, csplit, dbg, s); break;
        case 3: rc = launch_scatter<3>(head, head_stride, frustum, cam_mats, g, w, cells_total, lds, csplit, dbg, s); break;
        case 4: rc = launch_scatter<4>(head, head_stride, frustum, cam_mats, g, w, cells_total, lds, csplit, dbg, s); break;
    }
    if (rc) return rc;
    HEAL_LAUNCH_CHECK();
    return 0;
}

#ifdef HEAL_BUILD_EXPERIMENTAL   // measured negative at the scene level (profiles/r06_k4_shared_launch.json): not in the shipped library
extern "C" int heal_bev_pool_scatter_multi(int n_problems, const float* const* heads, const int32_t* head_strides,
                                           const float* const* frustums, const float* const* cam_mats, const int32_t* n_agents,
                                           const int32_t* n_cams, const int32_t* D, const int32_t* fH, const int32_t* fW,
                                           const int32_t* channels, const float* dx_host, const float* bx_host,
                                           const int32_t* nx_host, void* const* ws, const size_t* ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_problems >= 1 && n_problems <= LSS_MAX_PROBLEMS, "bev_pool_scatter_multi: 1..%d problems per launch", LSS_MAX_PROBLEMS);
    LssProblem pr[LSS_MAX_PROBLEMS];
    size_t lds = 0;
    const int n_dt = ceil_div(D[0], LSS_MT);
    for (int i = 0; i < n_problems; ++i) {
        HEAL_REQUIRE(n_agents[i] >= 1 && n_cams[i] >= 1 && D[i] >= 1 && fH[i] >= 1 && fW[i] >= 1, "bev_pool_pm: bad shape");
        HEAL_REQUIRE(fH[i] <= 64 && D[i] <= 64 && D[i] % 4 == 0, "bev_pool_pm: fH, D <= 64 and D %% 4 == 0 (got %d, %d)", fH[i], D[i]);
        HEAL_REQUIRE(ceil_div(D[i], LSS_MT) == n_dt, "bev_pool_scatter_multi: the problems of one launch must share ceil(D / 16)");
        HEAL_REQUIRE(head_strides[i] >= channels[i] + D[i] && head_strides[i] % 4 == 0 && ((uintptr_t)heads[i] & 15) == 0,
                     "bev_pool_pm: head rows must hold C + D floats, 16-B aligned (stride %d)", head_strides[i]);
        HEAL_REQUIRE(((uintptr_t)ws[i] & 255) == 0, "bev_pool_pm: workspace must be 256-B aligned");
        LssProblem& p = pr[i];
        if (pm_geometry(n_agents[i], channels[i], dx_host + 3 * i, bx_host + 3 * i, nx_host + 3 * i, p.g, p.cells_total)) return 1;
        p.g.n_agents = n_agents[i]; p.g.n_cams = n_cams[i]; p.g.D = D[i]; p.g.fH = fH[i]; p.g.fW = fW[i]; p.g.C = channels[i];
        const size_t l = (size_t)LssLds(fH[i], channels[i], n_dt).total * sizeof(float);
        HEAL_REQUIRE(l <= 150 * 1024, "bev_pool_pm: feature column of %zu B does not fit the LDS: use heal_bev_pool", l);
        lds = l > lds ? l : lds;
        Arena a(ws[i], ws_bytes[i]);
        HEAL_REQUIRE(carve_pm(a, channels[i], p.cells_total, p.ws), "bev_pool_pm: workspace too small (%zu < %zu)", ws_bytes[i], a.off);
        for (int j = 0; j < i; ++j) HEAL_REQUIRE(ws[j] != ws[i], "bev_pool_scatter_multi: every problem needs its own workspace");
        p.head = heads[i]; p.frustum = frustums[i]; p.cams = reinterpret_cast<const CamMats*>(cam_mats[i]); p.CT = head_strides[i];
        p.blocks = fW[i] * n_agents[i] * n_cams[i];
    }
    const int dbg = HEAL_DEBUG_ENV("HEAL_K4_DBG");
#define HEAL_LSS_PAIR(NDT_, P0, P1)                                                                                               \
    {                                                                                                                              \
        static bool attr_set = false;                                                                                              \
        if (!attr_set) {                                                                                                           \
            HEAL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lss_scatter_pair<NDT_>),                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                \
            attr_set = true;                                                                                                       \
        }                                                                                                                          \
        HEAL_LAUNCH_EV(k_lss_scatter_pair<NDT_>, dim3((P0).blocks + (P1).blocks), dim3(256 * NDT_), lds, s, P0, P1, dbg);         \
    }
    for (int i = 0; i + 1 < n_problems; i += 2) {
        switch (n_dt) {
            case 1: HEAL_LSS_PAIR(1, pr[i], pr[i + 1]) break;
            case 2: HEAL_LSS_PAIR(2, pr[i], pr[i + 1]) break;
            case 3: HEAL_LSS_PAIR(3, pr[i], pr[i + 1]) break;
            case 4: HEAL_LSS_PAIR(4, pr[i], pr[i + 1]) break;
            default: HEAL_REQUIRE(false, "bev_pool_pm: D must be <= 64");
        }
    }
#undef HEAL_LSS_PAIR
    if (n_problems & 1) {                        // an odd one out: the single-problem kernel
        const LssProblem& p = pr[n_problems - 1];
        int rc = 1;
        const size_t l1 = (size_t)LssLds(p.g.fH, p.g.C, n_dt).total * sizeof(float);
        switch (n_dt) {
            case 1: rc = launch_scatter<1>(p.head, p.CT, p.frustum, reinterpret_cast<const float*>(p.cams), p.g, p.ws, p.cells_total, l1, 1, dbg, s); break;
            case 2: rc = launch_scatter<2>(p.head, p.CT, p.frustum, reinterpret_cast<const float*>(p.cams), p.g, p.ws, p.cells_total, l1, 1, dbg, s); break;
            case 3: rc = launch_scatter<3>(p.head, p.CT, p.frustum, reinterpret_cast<const float*>(p.cams), p.g, p.ws, p.cells_total, l1, 1, dbg, s); break;
            case 4: rc = launch_scatter<4>(p.head, p.CT, p.frustum, reinterpret_cast<const float*>(p.cams), p.g, p.ws, p.cells_total, l1, 1, dbg, s); break;
        }
        if (rc) return rc;
    }
    HEAL_LAUNCH_CHECK();
    return 0;
}

#endif

extern "C" int heal_bev_pool_emit(int n_agents, int channels, const int32_t* nx_host, float* out, void* ws, size_t ws_bytes,
                                  void* stream) {
    LssGeom g;
    int cells_total = 0;
    if (pm_geometry(n_agents, channels, nullptr, nullptr, nx_host, g, cells_total)) return 1;
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)out & 15) == 0, "bev_pool_emit: misaligned workspace / output");
    Arena a(ws, ws_bytes);
    LssPmWs w;
    HEAL_REQUIRE(carve_pm(a, channels, cells_total, w), "bev_pool_emit: workspace too small (%zu < %zu)", ws_bytes, a.off);
    const int cells4 = g.nx[0] * g.nx[1] / 4;
    const dim3 grid(ceil_div(cells4, 64), channels / 16, n_agents * g.nx[2]);
    k_lss_canvas<<<grid, 256, 0, (hipStream_t)stream>>>(w, cells4, channels, reinterpret_cast<float4*>(out), cells_total);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_bev_pool_pm(const float* head, int head_stride, const float* frustum, const float* cam_mats,
                                int n_agents, int n_cams, int D, int fH, int fW, int channels, const float* dx_host,
                                const float* bx_host, const int32_t* nx_host, float* out, void* ws, size_t ws_bytes,
                                void* stream) {
    if (heal_bev_pool_scatter(head, head_stride, frustum, cam_mats, n_agents, n_cams, D, fH, fW, channels, dx_host, bx_host,
                              nx_host, ws, ws_bytes, stream))
        return 1;
    return heal_bev_pool_emit(n_agents, channels, nx_host, out, ws, ws_bytes, stream);
}

extern "C" int heal_bev_stem_block(int n_agents, int channels, const int32_t* nx_host, const float* w_main,
                                   const float* b_main, const float* w_down, const float* b_down, float* out_main,
                                   float* out_identity, void* ws, size_t ws_bytes, void* stream) {
    LssGeom g;
    int cells_total = 0;
    if (pm_geometry(n_agents, channels, nullptr, nullptr, nx_host, g, cells_total)) return 1;
    HEAL_REQUIRE(g.nx[2] == 1, "bev_stem_block: the pooled map must have one z bin (got %d)", g.nx[2]);
    HEAL_REQUIRE(channels % 32 == 0, "bev_stem_block: channels must be a multiple of 32 (got %d)", channels);
    HEAL_REQUIRE(w_main && w_down && out_main && out_identity, "bev_stem_block: null pointer");
    const int Ho = (g.nx[1] - 1) / 2 + 1, Wo = (g.nx[0] - 1) / 2 + 1;
    HEAL_REQUIRE(Wo % 4 == 0, "bev_stem_block: output width must be a multiple of 4 (got %d)", Wo);
    HEAL_REQUIRE((((uintptr_t)w_main | (uintptr_t)w_down | (uintptr_t)out_main | (uintptr_t)out_identity) & 15) == 0 &&
                     ((uintptr_t)ws & 255) == 0, "bev_stem_block: 16-B alignment");
    Arena a(ws, ws_bytes);
    LssPmWs w;
    HEAL_REQUIRE(carve_pm(a, channels, cells_total, w), "bev_stem_block: workspace too small (%zu < %zu)", ws_bytes, a.off);
    const int dbg = HEAL_DEBUG_ENV("HEAL_K4_DBG");   // timing experiments only (bits skip parts of the work: results invalid)
    const int tiles_x = ceil_div(Wo, 64);
    const int64_t blocks = (int64_t)n_agents * Ho * tiles_x;
    HEAL_REQUIRE(blocks < (1ll << 31), "bev_stem_block: grid too large");
    if (channels % 64 == 0)
        HEAL_LAUNCH_EV(k_bev_stem<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, channels, g.nx[0], g.nx[1],
                       cells_total, w_main, b_main, w_down, b_down, out_main, out_identity, Ho, Wo, tiles_x, dbg);
    else
        HEAL_LAUNCH_EV(k_bev_stem<32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, channels, g.nx[0], g.nx[1],
                       cells_total, w_main, b_main, w_down, b_down, out_main, out_identity, Ho, Wo, tiles_x, dbg);
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
        g.rdx[k] = 1.f / dx_host[k];
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
