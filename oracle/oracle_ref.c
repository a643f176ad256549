/*
 * oracle_ref.c -- TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the integer / index /
 * geometry parts of HEAL's perception hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (heal_amd/) never does.
 *
 * Parity status: the two algorithms here live in third-party packages that are NOT vendored in the
 * reference tree and are not installed in the build container:
 *   - voxelisation  -> spconv (unpinned; 1.2.1 VoxelGeneratorV2 or 2.x Point2VoxelCPU3d), called at
 *     opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:46-68.   PARITY UNPINNED: restated
 *     from the library's published algorithm and pinned by known-answer tests
 *     written for this build (tests/test_oracle_voxelize.py).
 *   - rotated IoU   -> shapely==2.0.0 / GEOS, called at opencood/utils/common_utils.py:230-270 from
 *     opencood/utils/box_utils.py:693-738 (nms_rotated).  PARITY UNPINNED for the GEOS arithmetic;
 *     the control flow of nms_rotated is restated line by line, the geometry is a convex clip in
 *     fp64 checked against analytic cases and scipy (tests/test_oracle_nms.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/build.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Hard voxelisation, sequential first-come semantics.
 * points [n,4] f32; range[6], vsize[3] f32; outputs sized for `cap` = min(n, max_voxels) voxels.
 * Returns the number of voxels M.  coords rows are (batch_idx, z, y, x).
 * ------------------------------------------------------------------------------------------------*/
int oracle_voxelize(const float* points, int n, const float* range, const float* vsize,
                    int max_points, int max_voxels, int batch_idx, float* voxels, int32_t* coords,
                    int32_t* num_points) {
    int grid[3];
    for (int j = 0; j < 3; ++j)
        grid[j] = (int)rint(((double)range[3 + j] - (double)range[j]) / (double)vsize[j]);
    const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
    int32_t* cell_to_voxel = (int32_t*)malloc(sizeof(int32_t) * (size_t)cells);
    if (!cell_to_voxel) return -1;
    memset(cell_to_voxel, 0xFF, sizeof(int32_t) * (size_t)cells); /* -1 */
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        int c[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            /* float arithmetic exactly as the library: (p - min) / size, floor */
            const float v = floorf((points[i * 4 + j] - range[j]) / vsize[j]);
            if (!(v >= 0.0f && v < (float)grid[j])) { failed = 1; break; }
            c[j] = (int)v;
        }
        if (failed) continue;
        const int64_t cell = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
        int vid = cell_to_voxel[cell];
        if (vid == -1) {
            if (voxel_num >= max_voxels) continue; /* new voxels past the cap are dropped */
            vid = voxel_num++;
            cell_to_voxel[cell] = vid;
            coords[vid * 4 + 0] = batch_idx;
            coords[vid * 4 + 1] = c[2];
            coords[vid * 4 + 2] = c[1];
            coords[vid * 4 + 3] = c[0];
            num_points[vid] = 0;
            memset(voxels + (size_t)vid * max_points * 4, 0, sizeof(float) * 4 * (size_t)max_points);
        }
        const int k = num_points[vid];
        if (k < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + k) * 4, points + (size_t)i * 4, sizeof(float) * 4);
            num_points[vid] = k + 1;
        }
    }
    free(cell_to_voxel);
    return voxel_num;
}

/* ------------------------------------------------------------------------------------------------
 * Convex quad intersection-over-union in fp64 on fp32 corners (common_utils.py:230-251:
 * Polygon(corners[0:4,:2]); intersection().area / union().area; cast to float32).
 * ------------------------------------------------------------------------------------------------*/
static double poly_area(const double* p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
    }
    return 0.5 * a;
}

/* Sutherland-Hodgman: clip `subj` (ns vertices) by the half plane left of edge a->b. */
static int clip_edge(const double* subj, int ns, double ax, double ay, double bx, double by, double* out) {
    int no = 0;
    const double ex = bx - ax, ey = by - ay;
    for (int i = 0; i < ns; ++i) {
        const int j = (i + 1 == ns) ? 0 : i + 1;
        const double px = subj[2 * i], py = subj[2 * i + 1];
        const double qx = subj[2 * j], qy = subj[2 * j + 1];
        const double dp = ex * (py - ay) - ey * (px - ax);
        const double dq = ex * (qy - ay) - ey * (qx - ax);
        const int pin = dp >= 0.0, qin = dq >= 0.0;
        if (pin) { out[2 * no] = px; out[2 * no + 1] = py; ++no; }
        if (pin != qin) {
            const double t = dp / (dp - dq);
            out[2 * no] = px + t * (qx - px);
            out[2 * no + 1] = py + t * (qy - py);
            ++no;
        }
    }
    return no;
}

float oracle_quad_iou(const float* qa, const float* qb) {
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = (double)qa[i]; b[i] = (double)qb[i]; }
    double sa = poly_area(a, 4), sb = poly_area(b, 4);
    if (sa < 0.0) { /* make both counter-clockwise */
        for (int i = 0; i < 2; ++i) {
            double tx = a[2 * i], ty = a[2 * i + 1];
            a[2 * i] = a[2 * (3 - i)]; a[2 * i + 1] = a[2 * (3 - i) + 1];
            a[2 * (3 - i)] = tx; a[2 * (3 - i) + 1] = ty;
        }
        sa = -sa;
    }
    if (sb < 0.0) {
        for (int i = 0; i < 2; ++i) {
            double tx = b[2 * i], ty = b[2 * i + 1];
            b[2 * i] = b[2 * (3 - i)]; b[2 * i + 1] = b[2 * (3 - i) + 1];
            b[2 * (3 - i)] = tx; b[2 * (3 - i) + 1] = ty;
        }
        sb = -sb;
    }
    double buf0[32], buf1[32];
    memcpy(buf0, a, sizeof(a));
    int n = 4;
    double* cur = buf0;
    double* nxt = buf1;
    for (int e = 0; e < 4 && n > 0; ++e) {
        const int f = (e + 1) & 3;
        n = clip_edge(cur, n, b[2 * e], b[2 * e + 1], b[2 * f], b[2 * f + 1], nxt);
        double* t = cur; cur = nxt; nxt = t;
    }
    double inter = (n >= 3) ? poly_area(cur, n) : 0.0;
    if (inter < 0.0) inter = 0.0;
    const double uni = sa + sb - inter;
    return (float)(inter / uni); /* 0/0 -> NaN, like shapely on degenerate boxes */
}

void oracle_quad_iou_matrix(const float* a, int n, const float* b, int m, float* iou) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) iou[(size_t)i * m + j] = oracle_quad_iou(a + 8 * i, b + 8 * j);
}

/* ------------------------------------------------------------------------------------------------
 * nms_rotated (box_utils.py:693-738): top `top` boxes by score (descending), greedy; a box is
 * removed when iou(picked, box) > thr (fp32 compare; NaN never removes).
 * `order` [k] = candidate indices already sorted by descending score (k <= top).
 * quads [N,4,2].  keep receives the picked original indices; returns how many.
 * ------------------------------------------------------------------------------------------------*/
int oracle_nms_rotated(const float* quads, const int32_t* order, int k, float thr, int32_t* keep) {
    uint8_t* dead = (uint8_t*)calloc((size_t)(k > 0 ? k : 1), 1);
    int nkeep = 0;
    for (int i = 0; i < k; ++i) {
        if (dead[i]) continue;
        const int bi = order[i];
        keep[nkeep++] = bi;
        for (int j = i + 1; j < k; ++j) {
            if (dead[j]) continue;
            const float v = oracle_quad_iou(quads + 8 * (size_t)bi, quads + 8 * (size_t)order[j]);
            if (v > thr) dead[j] = 1;
        }
    }
    free(dead);
    return nkeep;
}
