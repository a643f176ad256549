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
 *     written for this build (tests/test_host_cpu.py::test_oracle_voxelize_*).
 *   - rotated IoU   -> shapely==2.0.0 / GEOS, called at opencood/utils/common_utils.py:230-270 from
 *     opencood/utils/box_utils.py:693-738 (nms_rotated).  PARITY UNPINNED for the GEOS arithmetic;
 *     the control flow of nms_rotated is restated line by line, the geometry is a convex clip in
 *     fp64 checked against analytic cases and scipy (tests/test_host_cpu.py::test_oracle_quad_iou_*,
 *     test_oracle_nms_control_flow).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/cref.py).
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

/* =====================================================================================================
 * pcdet rotated BEV IoU / NMS.  fp32 throughout, MARGIN-inflated corner test, centroid
 * angle sort -- the arithmetic of opencood/pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp:38-230 (CPU twin of
 * iou3d_nms_kernel.cu:35-234), restated with precomputed polar angles (same comparisons, same swaps).
 * PINNED: tests compare this bit for bit with oracle/_ref/libpcdet_iou_ref.so, which is the reference's
 * own iou3d_cpu.cpp compiled where it lies (oracle/Makefile.ref).
 * Boxes are [x, y, z, dx, dy, dz, heading].
 * ===================================================================================================== */
#define PC_EPS 1e-8f
#define PC_MARGIN 1e-2f

typedef struct { float x, y; } pc_pt;

static float pc_cross3(pc_pt p1, pc_pt p2, pc_pt p0) { /* iou3d_cpu.cpp:63-65 */
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static void pc_corners(const float* box, pc_pt* c /*[5]*/) { /* iou3d_cpu.cpp:134-165, 119-123 */
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const float cs = cosf(box[6]), sn = sinf(box[6]);
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
    for (int k = 0; k < 4; ++k) {
        c[k].x = (px[k] - box[0]) * cs + (py[k] - box[1]) * (-sn) + box[0];
        c[k].y = (px[k] - box[0]) * sn + (py[k] - box[1]) * cs + box[1];
    }
    c[4] = c[0];
}

static int pc_inside(const float* box, pc_pt p) { /* iou3d_cpu.cpp:76-87 */
    const float cs = cosf(-box[6]), sn = sinf(-box[6]);
    const float rx = (p.x - box[0]) * cs + (p.y - box[1]) * (-sn);
    const float ry = (p.x - box[0]) * sn + (p.y - box[1]) * cs;
    return fabsf(rx) < box[3] / 2 + PC_MARGIN && fabsf(ry) < box[4] / 2 + PC_MARGIN;
}

/* segment p0->p1 against q0->q1; iou3d_cpu.cpp:89-117 */
static int pc_segment_hit(pc_pt p1, pc_pt p0, pc_pt q1, pc_pt q0, pc_pt* ans) {
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return 0;
    const float s1 = pc_cross3(q0, p1, p0), s2 = pc_cross3(p1, q1, p0);
    const float s3 = pc_cross3(p0, q1, q0), s4 = pc_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = pc_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > PC_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

float oracle_pcdet_overlap(const float* box_a, const float* box_b) { /* iou3d_cpu.cpp:129-221 */
    pc_pt ca[5], cb[5], pts[24];
    pc_corners(box_a, ca);
    pc_corners(box_b, cb);
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (pc_segment_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
                sx = sx + pts[cnt].x; sy = sy + pts[cnt].y; ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (pc_inside(box_a, cb[k])) { sx = sx + cb[k].x; sy = sy + cb[k].y; pts[cnt++] = cb[k]; }
        if (pc_inside(box_b, ca[k])) { sx = sx + ca[k].x; sy = sy + ca[k].y; pts[cnt++] = ca[k]; }
    }
    sx /= cnt; sy /= cnt; /* cnt == 0 -> NaN centre, no sort, no area: overlap 0 (as the reference) */
    float ang[24];
    for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - sy, pts[k].x - sx);
    for (int j = 0; j < cnt - 1; ++j)           /* bubble sort, swap when angle[i] > angle[i+1] (:201-210) */
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                pc_pt t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
                float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
        const float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
        area += ax * by - ay * bx;
    }
    return fabsf(area) / 2.0f;
}

float oracle_pcdet_iou_bev(const float* a, const float* b) { /* iou3d_cpu.cpp:223-230 */
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = oracle_pcdet_overlap(a, b);
    return so / fmaxf(sa + sb - so, PC_EPS);
}

float oracle_pcdet_iou_normal(const float* a, const float* b) { /* iou3d_nms_kernel.cu:314-325 (axis-aligned) */
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, PC_EPS);
}

/* mode 0: overlap area, 1: rotated IoU, 2: axis-aligned IoU.  out[n*m] */
void oracle_pcdet_matrix(const float* a, int n, const float* b, int m, int mode, float* out) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            const float* pa = a + 7 * (size_t)i; const float* pb = b + 7 * (size_t)j;
            out[(size_t)i * m + j] = mode == 0 ? oracle_pcdet_overlap(pa, pb)
                                   : mode == 1 ? oracle_pcdet_iou_bev(pa, pb) : oracle_pcdet_iou_normal(pa, pb);
        }
}

/* greedy NMS over boxes ALREADY in descending-score order: box i survives unless an earlier survivor j has
 * IoU(j, i) > thr (row = the earlier box: iou3d_nms_kernel.cu:293-307; host pass iou3d_nms.cpp:109-122).
 * rotated != 0 -> iou_bev, else iou_normal.  Returns the number kept, indices in keep[]. */
int oracle_pcdet_nms(const float* boxes, int n, float thr, int rotated, int64_t* keep) {
    char* dead = (char*)calloc((size_t)n + 1, 1);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[k++] = i;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float v = rotated ? oracle_pcdet_iou_bev(boxes + 7 * (size_t)i, boxes + 7 * (size_t)j)
                                    : oracle_pcdet_iou_normal(boxes + 7 * (size_t)i, boxes + 7 * (size_t)j);
            if (v > thr) dead[j] = 1;
        }
    }
    free(dead);
    return k;
}
