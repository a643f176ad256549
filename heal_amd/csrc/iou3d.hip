// pcdet rotated-BEV box ops on gfx950: overlap / IoU matrices and greedy NMS with pcdet's fp32,
// MARGIN-inflated semantics.  Replaces opencood/pcdet_utils/iou3d_nms/src/iou3d_nms_kernel.cu (+ the host-side
// mask walk of iou3d_nms.cpp:74-125, which here stays on the device: no mask D2H, no cudaMalloc per call).
//
//   k_bev_matrix   one thread per (a, b) pair: overlap area | rotated IoU | axis-aligned IoU
//   k_bev_nms_mask 64x64 tiles of the upper-triangular suppression matrix; a block is 4 waves, wave w tests
//                  every row of the tile against columns [16w, 16w+16): 4x shorter serial chain than a lane per row
//   k_bev_nms_walk one block: per 64-box group the first wave resolves the diagonal tile with lane broadcasts,
//                  then all threads OR the surviving rows into the removed words of the later groups
//
// Arithmetic: same operations in the same order as the reference's box_overlap (fp32, cosf/sinf/atan2f), so results
// agree with the CPU twin up to libm ulps; `-ffp-contract=off` keeps the products and sums unfused.
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

struct P2 { float x, y; };

__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ void bev_corners(const float* __restrict__ b, P2* c) {
    const float hx = b[3] / 2, hy = b[4] / 2;
    const float x1 = b[0] - hx, y1 = b[1] - hy, x2 = b[0] + hx, y2 = b[1] + hy;
    const float cs = cosf(b[6]), sn = sinf(b[6]);
    const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k].x = (px[k] - b[0]) * cs + (py[k] - b[1]) * (-sn) + b[0];
        c[k].y = (px[k] - b[0]) * sn + (py[k] - b[1]) * cs + b[1];
    }
    c[4] = c[0];
}

__device__ __forceinline__ bool bev_inside(const float* __restrict__ b, float cs, float sn, P2 p) {
    // cs/sn = cosf(-heading), sinf(-heading), hoisted by the caller (one evaluation per box instead of four)
    const float rx = (p.x - b[0]) * cs + (p.y - b[1]) * (-sn);
    const float ry = (p.x - b[0]) * sn + (p.y - b[1]) * cs;
    return fabsf(rx) < b[3] / 2 + 1e-2f && fabsf(ry) < b[4] / 2 + 1e-2f;
}

__device__ __forceinline__ bool segment_hit(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
          fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return false;
    const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// Overlap area of two rotated rectangles [x, y, z, dx, dy, dz, heading].
__device__ float bev_overlap(const float* __restrict__ a, const float* __restrict__ b) {
    // Disjoint bounding circles (with slack far above the 1e-2 corner margin): no crossing, no contained corner ->
    // the reference's point count is 0 and its area loop is empty: 0.  NaN/Inf inputs fail the test and take the
    // full path.
    {
        const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]), rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
        const float dx = a[0] - b[0], dy = a[1] - b[1], r = ra + rb + 0.25f;
        if (dx * dx + dy * dy > r * r * 1.0001f) return 0.f;
    }
    P2 ca[5], cb[5], pts[16];
    float ang[16];
    bev_corners(a, ca);
    bev_corners(b, cb);
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 h;
            if (segment_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], h) && cnt < 16) {
                sx = sx + h.x; sy = sy + h.y; pts[cnt++] = h;
            }
        }
    const float csa = cosf(-a[6]), sna = sinf(-a[6]), csb = cosf(-b[6]), snb = sinf(-b[6]);
    for (int k = 0; k < 4; ++k) {
        if (bev_inside(a, csa, sna, cb[k]) && cnt < 16) { sx = sx + cb[k].x; sy = sy + cb[k].y; pts[cnt++] = cb[k]; }
        if (bev_inside(b, csb, snb, ca[k]) && cnt < 16) { sx = sx + ca[k].x; sy = sy + ca[k].y; pts[cnt++] = ca[k]; }
    }
    sx /= (float)cnt; sy /= (float)cnt;
    for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - sy, pts[k].x - sx);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
                const float u = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = u;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
        const float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
        area += ax * by - ay * bx;
    }
    return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float bev_iou(const float* __restrict__ a, const float* __restrict__ b) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = bev_overlap(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

__device__ __forceinline__ float bev_iou_normal(const float* __restrict__ a, const float* __restrict__ b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

__global__ __launch_bounds__(256) void k_bev_matrix(const float* __restrict__ A, int n, const float* __restrict__ B,
                                                   int m, int mode, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n * m) return;
    const int i = (int)(t / m), j = (int)(t - (long long)i * m);
    float a[7], b[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { a[k] = A[(size_t)i * 7 + k]; b[k] = B[(size_t)j * 7 + k]; }
    out[t] = mode == 0 ? bev_overlap(a, b) : mode == 1 ? bev_iou(a, b) : bev_iou_normal(a, b);
}

// mask[row * W + cb] bit c: IoU(row, cb*64 + c) > thr, for cb >= row/64 (and c > row%64 on the diagonal tile)
__global__ __launch_bounds__(256) void k_bev_nms_mask(const float* __restrict__ boxes, int n, float thr, int rotated,
                                                     int W, unsigned long long* __restrict__ mask) {
    // linear block id -> upper-triangular tile (rb <= cb)
    int rb = 0, rem = blockIdx.x;
    while (rem >= W - rb) { rem -= W - rb; ++rb; }
    const int cb = rb + rem;
    __shared__ float colb[64 * 7];
    __shared__ unsigned short part[64][4];
    for (int e = threadIdx.x; e < 64 * 7; e += 256) {
        const int g = cb * 64 * 7 + e;
        colb[e] = g < n * 7 ? boxes[g] : 0.f;
    }
    __syncthreads();
    const int r = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int row = rb * 64 + r;
    unsigned bits = 0;
    if (row < n) {
        float a[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) a[k] = boxes[(size_t)row * 7 + k];
        const int csize = min(64, n - cb * 64);
        for (int c = chunk * 16; c < chunk * 16 + 16; ++c) {
            if (c >= csize) break;
            if (rb == cb && c <= r) continue;
            const float v = rotated ? bev_iou(a, colb + c * 7) : bev_iou_normal(a, colb + c * 7);
            if (v > thr) bits |= 1u << (c - chunk * 16);
        }
    }
    part[r][chunk] = (unsigned short)bits;
    __syncthreads();
    if (threadIdx.x < 64 && row < n) {
        const unsigned long long w = (unsigned long long)part[r][0] | ((unsigned long long)part[r][1] << 16) |
                                     ((unsigned long long)part[r][2] << 32) | ((unsigned long long)part[r][3] << 48);
        mask[(size_t)row * W + cb] = w;
    }
}

__global__ __launch_bounds__(256) void k_bev_nms_walk(const unsigned long long* __restrict__ mask, int n, int W,
                                                     unsigned long long* __restrict__ removed /*[W] scratch*/,
                                                     long long* __restrict__ keep, int* __restrict__ num_keep) {
    __shared__ unsigned long long alive_s;
    __shared__ int base_s;
    for (int j = threadIdx.x; j < W; j += 256) removed[j] = 0ull;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int blk = 0; blk < W; ++blk) {
        if (threadIdx.x < 64) {
            const int l = threadIdx.x, r = blk * 64 + l;
            const unsigned long long diag = r < n ? mask[(size_t)r * W + blk] : 0ull;
            unsigned long long rem = removed[blk];
            unsigned long long alive = 0ull;
            const int cnt = min(64, n - blk * 64);
            for (int i = 0; i < cnt; ++i) {
                const unsigned long long di = __shfl(diag, i, 64);
                if (!((rem >> i) & 1ull)) { alive |= 1ull << i; rem |= di; }
            }
            // survivors of this group, in index order
            const bool kept = (alive >> l) & 1ull;
            const int pos = base_s + __popcll(alive & lanemask_lt());
            if (kept) keep[pos] = r;
            if (l == 0) alive_s = alive;
        }
        __syncthreads();
        const unsigned long long alive = alive_s;
        if (threadIdx.x == 0) base_s += __popcll(alive);
        for (int j = blk + 1 + threadIdx.x; j < W; j += 256) {
            unsigned long long acc = removed[j];
            unsigned long long a = alive;
            while (a) {
                const int i = __ffsll((long long)a) - 1;
                a &= a - 1;
                acc |= mask[(size_t)(blk * 64 + i) * W + j];
            }
            removed[j] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_keep = base_s;
}

// shared with heal_nms_quads (decode_nms.hip): greedy walk over an upper-triangular suppression bit matrix
int launch_bev_nms_walk(const unsigned long long* mask, int n, int W, unsigned long long* removed, long long* keep,
                        int* num_keep, hipStream_t s) {
    k_bev_nms_walk<<<1, 256, 0, s>>>(mask, n, W, removed, keep, num_keep);
    HEAL_LAUNCH_CHECK();
    return 0;
}

}  // namespace heal

using namespace heal;

extern "C" int heal_boxes_bev_matrix(const float* boxes_a, int n, const float* boxes_b, int m, int mode, float* out,
                                     void* stream) {
    HEAL_REQUIRE(n >= 0 && m >= 0, "heal_boxes_bev_matrix: negative box count");
    HEAL_REQUIRE(mode >= 0 && mode <= 2, "heal_boxes_bev_matrix: mode must be 0 (overlap), 1 (iou) or 2 (iou_normal)");
    if (n == 0 || m == 0) return 0;
    HEAL_REQUIRE(boxes_a && boxes_b && out, "heal_boxes_bev_matrix: null pointer");
    const long long total = (long long)n * m;
    k_bev_matrix<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(boxes_a, n, boxes_b, m, mode, out);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t heal_nms_bev_workspace(int n) {
    if (n <= 0) return 256;
    const size_t W = ((size_t)n + 63) / 64;
    return ((size_t)n * W + W) * sizeof(unsigned long long) + 256;
}

extern "C" int heal_nms_bev(const float* boxes_sorted, int n, float thresh, int rotated, void* workspace,
                            size_t workspace_bytes, long long* keep, int* num_keep, void* stream) {
    HEAL_REQUIRE(n >= 0, "heal_nms_bev: negative box count");
    HEAL_REQUIRE(num_keep != nullptr, "heal_nms_bev: null num_keep");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        HEAL_FILL(num_keep, 0, sizeof(int), st);
        return 0;
    }
    HEAL_REQUIRE(boxes_sorted && keep && workspace, "heal_nms_bev: null pointer");
    HEAL_REQUIRE(workspace_bytes >= heal_nms_bev_workspace(n), "heal_nms_bev: workspace too small");
    const int W = (n + 63) / 64;
    Arena ar(workspace, workspace_bytes);
    unsigned long long* mask = ar.take<unsigned long long>((size_t)n * W);
    unsigned long long* removed = ar.take<unsigned long long>(W);
    const long long tiles = (long long)W * (W + 1) / 2;
    k_bev_nms_mask<<<(unsigned)tiles, 256, 0, st>>>(boxes_sorted, n, thresh, rotated, W, mask);
    k_bev_nms_walk<<<1, 256, 0, st>>>(mask, n, W, removed, keep, num_keep);
    HEAL_LAUNCH_CHECK();
    return 0;
}
