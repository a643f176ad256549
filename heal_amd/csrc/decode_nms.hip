// K8 -- anchor box decode, score / size / z filters, rotated NMS and range mask, all on device.
//
// Reference arithmetic:
//   opencood/data_utils/post_processor/voxel_postprocessor.py:245-405 (post_process),
//   :407-453 (delta_to_boxes3d); opencood/utils/common_utils.py:104-113 (limit_period), :139-161
//   (rotate_points_along_z); opencood/utils/box_utils.py:152-204 (boxes_to_corners_3d 'hwl'),
//   :278-316 (project_box3d), :840-890 (size / z filters), :693-738 (nms_rotated: top-1000 by
//   score, greedy, suppress iou > thr), :384-421 (mask_boxes_outside_range_numpy);
//   IoU = shapely Polygon(corners[0:4,:2]) intersection/union in fp64 -> fp32
//   (opencood/utils/common_utils.py:230-270).
// The reference moves the candidates to the host and runs O(K^2) Python->GEOS calls.  Here:
//   k_decode_key    one thread per anchor: sigmoid, decode, filters -> survivors compacted into a list of 64-bit composites
//                   (score bits | anchor index: distinct; descending composite order = the reference's order incl. ties)
//   k_rank_prepare  rank of every candidate by counting (multi-block, list in LDS) and re-decode of the top-k straight into
//                   slot `rank` (cheaper than storing 96 B for each of 131 072 anchors); a radix select + bitonic sort by
//                   one block only when more than 4096 anchors pass the threshold
//   k_nms_mask      64x64 tiles of the upper-triangular suppression bit matrix; fp64 convex clip with the polygons in LDS
//   k_nms_reduce    the matrix copied to LDS; one wave resolves each 64-row diagonal tile with a scalar bit chain and ORs the
//                   surviving rows into the removed set; ordered compaction of the output by the whole block
#include <string.h>
#include "prims.h"
#include "../../include/heal_amd.h"

namespace heal {

struct DecodeParams {
    int H, W, A, num_bins;
    float score_thr, dir_offset, period, two_pi;
    float tfm[16];
    float gt_range[6];
    uint32_t key_base;  // keys are bits(score) - key_base (>= 1 for candidates)
};

struct Box3D {
    float c[8][3];
    float score;
    bool pass;
};

__device__ __forceinline__ float limit_period_f(float val, float offset, float period) {
    return val - floorf(val / period + offset) * period;
}

// Full per-anchor pipeline up to (and including) the size / z filters.
__device__ __forceinline__ void decode_anchor(const float* __restrict__ cls, const float* __restrict__ reg,
                                              const float* __restrict__ dir,
                                              const float* __restrict__ anchors, const DecodeParams& p,
                                              int j, bool want_corners, Box3D& o) {
    const int HW = p.H * p.W;
    const int a = j % p.A;
    const int hw = j / p.A;
    const float logit = cls[(size_t)a * HW + hw];
    const float prob = 1.f / (1.f + expf(-logit));
    o.score = prob;
    o.pass = false;
    if (!(prob > p.score_thr)) return;
    float d[7], an[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        d[k] = reg[(size_t)(a * 7 + k) * HW + hw];
        an[k] = anchors[(size_t)j * 7 + k];
    }
    const float ad = sqrtf(an[4] * an[4] + an[5] * an[5]);
    const float x = d[0] * ad + an[0];
    const float y = d[1] * ad + an[1];
    const float z = d[2] * an[3] + an[2];
    const float bh = expf(d[3]) * an[3];
    const float bw = expf(d[4]) * an[4];
    const float bl = expf(d[5]) * an[5];
    float yaw = d[6] + an[6];
    if (dir != nullptr) {
        int label = 0;
        float best = dir[(size_t)(a * p.num_bins) * HW + hw];
        for (int b = 1; b < p.num_bins; ++b) {
            const float v = dir[(size_t)(a * p.num_bins + b) * HW + hw];
            if (v > best) { best = v; label = b; }
        }
        const float dir_rot = limit_period_f(yaw - p.dir_offset, 0.f, p.period);
        yaw = dir_rot + p.dir_offset + p.period * (float)label;
        yaw = limit_period_f(yaw, 0.5f, p.two_pi);
    }
    // corners: dims (l, w, h) * template / 2, rotate about z, translate, project
    const float cosa = cosf(yaw), sina = sinf(yaw);
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1};
    const float sy[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
    const float sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float px = bl * (sx[k] / 2.f), py = bw * (sy[k] / 2.f), pz = bh * (sz[k] / 2.f);
        // [px py pz] @ [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]]
        const float rx = (px * cosa + py * (-sina)) + pz * 0.f;
        const float ry = (px * sina + py * cosa) + pz * 0.f;
        const float rz = (px * 0.f + py * 0.f) + pz * 1.f;
        const float cx = rx + x, cy = ry + y, cz = rz + z;
        const float qx = ((p.tfm[0] * cx + p.tfm[1] * cy) + p.tfm[2] * cz) + p.tfm[3];
        const float qy = ((p.tfm[4] * cx + p.tfm[5] * cy) + p.tfm[6] * cz) + p.tfm[7];
        const float qz = ((p.tfm[8] * cx + p.tfm[9] * cy) + p.tfm[10] * cz) + p.tfm[11];
        if (want_corners) { o.c[k][0] = qx; o.c[k][1] = qy; o.c[k][2] = qz; }
        xmin = fminf(xmin, qx); xmax = fmaxf(xmax, qx);
        ymin = fminf(ymin, qy); ymax = fmaxf(ymax, qy);
        zmin = fminf(zmin, qz); zmax = fmaxf(zmax, qz);
    }
    const float x_len = xmax - xmin, y_len = ymax - ymin;
    // remove_large_pred_bbx: x_len<=6 & y_len<=6 & bool(z_len), where the reference derives z_len
    // from the y column (box_utils.py:862-867); remove_bbx_abnormal_z: z in [-3, 1]
    o.pass = (x_len <= 6.f) && (y_len <= 6.f) && (y_len != 0.f) && (zmin >= -3.f) && (zmax <= 1.f);
}

// One thread per anchor: decode + filters.  Survivors are COMPACTED into a candidate list of 64-bit composites
// (score key << 32 | anchor index; one wave-aggregated atomic per wave): all composites are distinct, and descending
// composite order = the reference's order (descending score, ties: larger anchor index first == stable argsort reversed), so
// the order in which the atomics hand out slots does not matter.
__global__ __launch_bounds__(256) void k_decode_key(const float* __restrict__ cls,
                                                   const float* __restrict__ reg,
                                                   const float* __restrict__ dir,
                                                   const float* __restrict__ anchors, DecodeParams p,
                                                   int n, unsigned long long* __restrict__ cand_list,
                                                   int* __restrict__ n_cand) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    bool cand = false;
    unsigned long long comp = 0ull;
    if (j < n) {
        Box3D b;
        decode_anchor(cls, reg, dir, anchors, p, j, false, b);
        cand = b.pass;
        comp = ((unsigned long long)(__float_as_uint(b.score) - p.key_base) << 32) | (unsigned)j;
    }
    const unsigned long long m = __ballot(cand);
    if (!m) return;
    int base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(n_cand, __popcll(m));
    base = __shfl(base, 0, 64);
    if (cand) cand_list[base + __popcll(m & lanemask_lt())] = comp;
}

// Fallback for more than TOPK_CAP candidates (every anchor above the threshold: not a frame a detector produces): ONE block finds
// the top `top` of the candidate list -- an MSB-first radix select (8-bit digits over the 64-bit composite, histogram in LDS, the
// list re-read from global memory per pass) finds the composite T of rank `top`; the candidates >= T are exactly the top `top`
// (composites are distinct) -- and sorts them descending with a bitonic network: sbuf[0 .. K) afterwards, K = min(N, top).
constexpr int TOPK_CAP = 4096, TOPK_THREADS = 256;
__device__ void topk_select_sort(const unsigned long long* __restrict__ cand_list, int N, int top,
                                 unsigned long long* sbuf /* LDS [TOPK_CAP] */, int* hist /* LDS [256] */,
                                 unsigned long long* s_prefix, int* s_need, int* s_count) {
    const int K = min(N, top), nthr = blockDim.x;
    // radix select: after the loop, elements with (v >> shift) > prefix are in, == prefix are the ties to split further
    if (threadIdx.x == 0) { *s_prefix = 0ull; *s_need = K; }
    __syncthreads();
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += nthr) hist[i] = 0;
        __syncthreads();
        const unsigned long long prefix = *s_prefix;
        const int hi_shift = shift + 8;
        for (int i = threadIdx.x; i < N; i += nthr) {
            const unsigned long long v = cand_list[i];
            const bool in_prefix = hi_shift >= 64 ? true : ((v >> hi_shift) == prefix);
            if (in_prefix) atomicAdd(&hist[(int)((v >> shift) & 0xFFull)], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int need = *s_need, d = 255;
            for (; d > 0; --d) {           // walk the digits from the top until `need` is covered
                if (hist[d] >= need) break;
                need -= hist[d];
            }
            *s_need = need;                 // still needed among the elements whose digit == d
            *s_prefix = (prefix << 8) | (unsigned long long)d;
        }
        __syncthreads();
    }
    // s_prefix is now the composite of rank K (need == 1): gather everything >= it
    const unsigned long long T = *s_prefix;
    if (threadIdx.x == 0) *s_count = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += nthr) {
        const unsigned long long v = cand_list[i];
        if (v >= T) { const int q = atomicAdd(s_count, 1); if (q < TOPK_CAP) sbuf[q] = v; }
    }
    __syncthreads();
    const int M = min(*s_count, TOPK_CAP);
    int P = 1;
    while (P < M) P <<= 1;
    for (int i = M + threadIdx.x; i < P; i += nthr) sbuf[i] = 0ull;   // pad: sorts to the end (descending)
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += nthr) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = sbuf[i], b = sbuf[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { sbuf[i] = b; sbuf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- fp64 convex quad IoU (same operation order as oracle/oracle_ref.c) ---------------------------
__device__ __forceinline__ double poly_area(const double* p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
    }
    return 0.5 * a;
}

__device__ __forceinline__ void make_ccw(double* q, double& area) {
    area = poly_area(q, 4);
    if (area < 0.0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double tx = q[2 * i], ty = q[2 * i + 1];
            q[2 * i] = q[2 * (3 - i)]; q[2 * i + 1] = q[2 * (3 - i) + 1];
            q[2 * (3 - i)] = tx; q[2 * (3 - i) + 1] = ty;
        }
        area = -area;
    }
}

// The same clip with the two polygon buffers in LDS.  A thread-private `double buf[32]` indexed by the running vertex count lives
// in SCRATCH memory: every vertex of every edge pass is then a dependent store -> load round trip to L2 (k_nms_mask: 30 us for
// two clips per thread).  Layout: element e (= 2 * vertex + coordinate, < 16: clipping a convex quad by four half planes
// leaves at most 8 vertices) of thread t at buf[e * NT + t] -- consecutive lanes, consecutive words, whatever e each lane is at.
template <int NT>
__device__ __forceinline__ int clip_edge_lds(const double* subj, int ns, double ax, double ay, double bx, double by, double* out) {
    int no = 0;
    const double ex = bx - ax, ey = by - ay;
    for (int i = 0; i < ns; ++i) {
        const int j = (i + 1 == ns) ? 0 : i + 1;
        const double px = subj[(2 * i) * NT], py = subj[(2 * i + 1) * NT];
        const double qx = subj[(2 * j) * NT], qy = subj[(2 * j + 1) * NT];
        const double dp = ex * (py - ay) - ey * (px - ax);
        const double dq = ex * (qy - ay) - ey * (qx - ax);
        const bool pin = dp >= 0.0, qin = dq >= 0.0;
        if (pin) {
            if (no < 8) { out[(2 * no) * NT] = px; out[(2 * no + 1) * NT] = py; }
            ++no;
        }
        if (pin != qin) {
            const double t = dp / (dp - dq);
            if (no < 8) { out[(2 * no) * NT] = px + t * (qx - px); out[(2 * no + 1) * NT] = py + t * (qy - py); }
            ++no;
        }
    }
    return no < 8 ? no : 8;
}

template <int NT>
__device__ __forceinline__ float quad_iou_lds(const float* qa, const float* qb, double* lds /* this thread's column of [32][NT] */) {
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (double)qa[i]; b[i] = (double)qb[i]; }
    double sa, sb;
    make_ccw(a, sa);
    make_ccw(b, sb);
    double* cur = lds;
    double* nxt = lds + 16 * NT;
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i * NT] = a[i];
    int n = 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = (e + 1) & 3;
        if (n > 0) {
            n = clip_edge_lds<NT>(cur, n, b[2 * e], b[2 * e + 1], b[2 * f], b[2 * f + 1], nxt);
            double* t = cur; cur = nxt; nxt = t;
        }
    }
    double inter = 0.0;
    if (n >= 3) {
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            inter += cur[(2 * i) * NT] * cur[(2 * j + 1) * NT] - cur[(2 * j) * NT] * cur[(2 * i + 1) * NT];
        }
        inter = 0.5 * inter;
    }
    if (inter < 0.0) inter = 0.0;
    const double uni = sa + sb - inter;
    return (float)(inter / uni);
}

// ---- top-k re-decode --------------------------------------------------------------------------------
struct NmsBufs {
    float* corners;   // [top][8][3]
    float* scores;    // [top]
    float* quads;     // [top][4][2]
    int* inrange;     // [top]
    unsigned long long* mask;  // [top][words]
    unsigned long long* diag;  // [words][64] the diagonal tiles once more, contiguous (scalar loads in k_nms_reduce), or NULL
    int* n_cand;
};

// candidate (anchor j) decoded into slot r of the top-k buffers: corners, BEV quad, score, inside-gt_range flag
__device__ __forceinline__ void decode_slot(const float* __restrict__ cls, const float* __restrict__ reg,
                                            const float* __restrict__ dir, const float* __restrict__ anchors,
                                            const DecodeParams& p, int j, int r, const NmsBufs& nb) {
    Box3D b;
    decode_anchor(cls, reg, dir, anchors, p, j, true, b);
    bool inside = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            nb.corners[((size_t)r * 8 + k) * 3 + ax] = b.c[k][ax];
            inside = inside && (b.c[k][ax] >= p.gt_range[ax]) && (b.c[k][ax] <= p.gt_range[3 + ax]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        nb.quads[(size_t)r * 8 + 2 * k] = b.c[k][0];
        nb.quads[(size_t)r * 8 + 2 * k + 1] = b.c[k][1];
    }
    nb.scores[r] = b.score;
    nb.inrange[r] = inside ? 1 : 0;
}

// Rank + re-decode of the top candidates in one multi-block launch.  Composites are distinct, so the rank of a candidate in
// descending order is the number of composites greater than its own: every block copies the candidate list (<= TOPK_CAP entries,
// a few hundred in practice) into LDS, four threads count a quarter of it each for one candidate, and the candidate is decoded
// straight into slot `rank` of the top-k buffers -- no sort, no `sel` list, no single-block kernel on the critical path
// (k_topk_sort 12 us + k_nms_prepare 5 us before).  More than TOPK_CAP candidates: block 0 alone runs topk_select_sort and decodes.
constexpr int RANK_PER_BLOCK = TOPK_THREADS / 4;
__global__ __launch_bounds__(TOPK_THREADS) void k_rank_prepare(const float* __restrict__ cls, const float* __restrict__ reg,
                                                              const float* __restrict__ dir,
                                                              const float* __restrict__ anchors, DecodeParams p, int n,
                                                              const unsigned long long* __restrict__ cand_list, int top,
                                                              NmsBufs nb) {
    __shared__ unsigned long long sc[TOPK_CAP];
    __shared__ int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_need, s_count;
    const int N = *nb.n_cand;
    int r = -1, j = 0;
    if (N <= TOPK_CAP) {
        if ((int)blockIdx.x * RANK_PER_BLOCK >= N) return;
        for (int i = threadIdx.x; i < N; i += TOPK_THREADS) sc[i] = cand_list[i];
        __syncthreads();
        const int c = blockIdx.x * RANK_PER_BLOCK + (threadIdx.x >> 2), q = threadIdx.x & 3;
        const unsigned long long mine = sc[min(c, N - 1)];
        int cnt = 0;
#pragma unroll 8
        for (int i = q; i < N; i += 4) cnt += sc[i] > mine ? 1 : 0;
        cnt += __shfl_xor(cnt, 1, 64);
        cnt += __shfl_xor(cnt, 2, 64);
        if (q == 0 && c < N && cnt < top) { r = cnt; j = (int)(unsigned)(mine & 0xFFFFFFFFull); }
    } else {
        if (blockIdx.x != 0) return;
        topk_select_sort(cand_list, N, top, sc, hist, &s_prefix, &s_need, &s_count);
        for (int rr = threadIdx.x; rr < min(N, top); rr += TOPK_THREADS)
            decode_slot(cls, reg, dir, anchors, p, (int)(unsigned)(sc[rr] & 0xFFFFFFFFull), rr, nb);
        return;
    }
    if (r >= 0) decode_slot(cls, reg, dir, anchors, p, j, r, nb);
}

// One 64x64 tile of the upper-triangular suppression bit matrix per block (8 waves).  Phase 1: all 4096 bounding-box tests
// (strictly separated boxes => empty intersection => IoU 0, or NaN for a degenerate pair: never above the threshold, exactly
// as the full clip would conclude), the surviving pairs compacted into an LDS list.  Phase 2: the threads take surviving
// pairs round-robin, one fp64 convex clip each -- the first version gave every lane 8 columns to walk serially, so a tile
// took as long as its unluckiest lane's chain of clips (95 us for 136 tiles).
constexpr int NMS_SPLIT = 8, NMS_COLS = 64 / NMS_SPLIT;
constexpr size_t NMS_POLY_LDS = (size_t)32 * 64 * NMS_SPLIT * sizeof(double);   // 128 KB: two 8-vertex fp64 polygons per thread
__global__ __launch_bounds__(64 * NMS_SPLIT) void k_nms_mask(NmsBufs nb, int top, int words, float thr) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int K = nb.n_cand ? min(*nb.n_cand, top) : top;
    if (bi * 64 >= K || bj * 64 >= K) return;
    __shared__ float rq[64][8], cq[64][8];
    __shared__ float rbb[64][4], cbb[64][4];  // axis-aligned bounds (xmin, xmax, ymin, ymax)
    __shared__ unsigned long long bits[64];
    __shared__ unsigned short pairs[4096];
    __shared__ int n_pairs;
    extern __shared__ __attribute__((aligned(16))) double s_poly[];   // [32][64 * NMS_SPLIT]: the clip's polygon buffers
    const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    if (chunk < 2) {
        const int idx = (chunk == 0 ? bi : bj) * 64 + lane;
        float (*q)[8] = chunk == 0 ? rq : cq;
        float (*bb)[4] = chunk == 0 ? rbb : cbb;
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        if (idx < K) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float px = nb.quads[(size_t)idx * 8 + 2 * k], py = nb.quads[(size_t)idx * 8 + 2 * k + 1];
                q[lane][2 * k] = px; q[lane][2 * k + 1] = py;
                x0 = fminf(x0, px); x1 = fmaxf(x1, px); y0 = fminf(y0, py); y1 = fmaxf(y1, py);
            }
        }
        bb[lane][0] = x0; bb[lane][1] = x1; bb[lane][2] = y0; bb[lane][3] = y1;
    }
    if (threadIdx.x < 64) bits[threadIdx.x] = 0ull;
    if (threadIdx.x == 0) n_pairs = 0;
    __syncthreads();
    {   // phase 1: row = lane, columns chunk*8 .. chunk*8+7
        const int i = bi * 64 + lane;
        const int jn = min(64, K - bj * 64);
        const float x0 = rbb[lane][0], x1 = rbb[lane][1], y0 = rbb[lane][2], y1 = rbb[lane][3];
#pragma unroll
        for (int t0 = 0; t0 < NMS_COLS; ++t0) {
            const int t = chunk * NMS_COLS + t0;
            bool live = i < K && t < jn && !(bi == bj && t <= lane) &&
                        !(cbb[t][0] > x1 || cbb[t][1] < x0 || cbb[t][2] > y1 || cbb[t][3] < y0);
            const unsigned long long m = __ballot(live);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&n_pairs, __popcll(m));
                base = __shfl(base, 0, 64);
                if (live) pairs[base + __popcll(m & lanemask_lt())] = (unsigned short)(lane * 64 + t);
            }
        }
    }
    __syncthreads();
    const int np = n_pairs;
    for (int e = threadIdx.x; e < np; e += 64 * NMS_SPLIT) {
        const int pr = pairs[e], r = pr >> 6, t = pr & 63;
        const float v = quad_iou_lds<64 * NMS_SPLIT>(rq[r], cq[t], s_poly + threadIdx.x);
        if (v > thr) atomicOr(&bits[r], 1ull << t);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int i = bi * 64 + threadIdx.x;
        if (i < K) nb.mask[(size_t)i * words + bj] = bits[threadIdx.x];
        if (bi == bj && nb.diag) nb.diag[(size_t)i] = i < K ? bits[threadIdx.x] : 0ull;
    }
}

// Greedy pass.  The upper-triangular tiles of the suppression matrix of the top-k (<= 1088 x 17 words = 148 KB) are first copied
// into LDS by the whole block (1024 threads, eight loads in flight each), then wave 0 walks the 64-row blocks:
//   * the removed set of the block's columns = OR over the KEPT rows of all earlier blocks of their word for this column: lane l
//     takes row l of every earlier block (independent LDS reads), one butterfly OR across the wave;
//   * the diagonal tile is resolved by a SCALAR bit chain: its 64 words arrive by scalar loads (k_nms_mask writes the diagonal
//     tiles a second time, contiguously), and a row costs a bit test, a select and an OR (bit i of the set is final once row i
//     has been looked at -- the tile is strictly upper triangular -- so alive = valid & ~set afterwards);
// (one wave doing a dependent vector chain per candidate + 64 predicated LDS reads per block and lane: 20-27 us for 600
// candidates).  The survivors inside gt_range are then compacted in pick order by the whole block.
constexpr int NMS_RED_THREADS = 1024, NMS_RED_MAXK = 1088;
__global__ __launch_bounds__(NMS_RED_THREADS) void k_nms_reduce(NmsBufs nb, const unsigned long long* __restrict__ diag,
                                                               int top, int words, float* __restrict__ out_corners,
                                                               float* __restrict__ out_scores, int* __restrict__ out_count,
                                                               int max_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long smask[];  // [K][W], upper-triangular tiles only
    __shared__ unsigned long long keep_bits[NMS_RED_MAXK / 64];
    __shared__ short s_row[NMS_RED_MAXK];   // pick-order list of the surviving rows inside gt_range
    __shared__ int s_kept;
    __shared__ unsigned char s_inr[NMS_RED_MAXK];
    const int K = min(*nb.n_cand, top);
    const int W = (K + 63) / 64;
    for (int r = threadIdx.x; r < K; r += NMS_RED_THREADS) s_inr[r] = (unsigned char)(nb.inrange[r] != 0);
    for (int e0 = threadIdx.x; e0 < K * W; e0 += NMS_RED_THREADS * 8) {     // eight loads in flight per thread, then the LDS stores
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + NMS_RED_THREADS * u, K * W - 1), r = e / W, c = e - r * W;
            v[u] = nb.mask[(size_t)r * words + max(c, r / 64)];           // unconditional, clamped into the written triangle
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + NMS_RED_THREADS * u < K * W) smask[e0 + NMS_RED_THREADS * u] = v[u];   // entries below the diagonal tile: never read
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        int base = 0;
        for (int blk = 0; blk < W; ++blk) {
            const int r = blk * 64 + l;
            // removed set of this block's columns: kept rows of the earlier blocks
            unsigned long long part = 0ull;
            for (int b = 0; b < blk; ++b) {
                const unsigned long long m = smask[(size_t)(b * 64 + l) * W + blk];   // row b * 64 + l < K: b < blk <= W - 1
                part |= ((keep_bits[b] >> l) & 1ull) ? m : 0ull;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part |= __shfl_xor(part, o, 64);
            unsigned long long rem = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(part >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)part);
            const int cnt = min(64, K - blk * 64);
            const unsigned long long valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            const unsigned long long* __restrict__ dg = diag + (size_t)blk * 64;   // uniform address: scalar loads
#pragma unroll
            for (int h = 0; h < 2; ++h) {           // 32 rows at a time: 64 SGPRs of diagonal words, loaded before the chain needs them
                unsigned long long d[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) d[i] = dg[h * 32 + i];
                // rem |= bit(rem, row) ? 0 : d[row] -- three scalar instructions per row (the compiler's form of the same expression
                // takes six, and turns the loads into conditional ones unless they are pinned in front of the chain)
                unsigned long long t;
#define HEAL_NMS_R(N, BIT) "s_bitcmp1_b64 %0, " #BIT "\n\ts_cselect_b64 %1, 0, %" #N "\n\ts_or_b64 %0, %0, %1\n\t"
#define HEAL_NMS_ROW8(I, B0, B1, B2, B3, B4, B5, B6, B7)                                                                  \
    asm volatile(HEAL_NMS_R(2, B0) HEAL_NMS_R(3, B1) HEAL_NMS_R(4, B2) HEAL_NMS_R(5, B3) HEAL_NMS_R(6, B4) HEAL_NMS_R(7, B5)  \
                 HEAL_NMS_R(8, B6) HEAL_NMS_R(9, B7)                                                                       \
                 : "+s"(rem), "=&s"(t)                                                                                     \
                 : "s"(d[I]), "s"(d[I + 1]), "s"(d[I + 2]), "s"(d[I + 3]), "s"(d[I + 4]), "s"(d[I + 5]), "s"(d[I + 6]),  \
                   "s"(d[I + 7])                                                                                           \
                 : "scc");
                if (h == 0) {
                    HEAL_NMS_ROW8(0, 0, 1, 2, 3, 4, 5, 6, 7) HEAL_NMS_ROW8(8, 8, 9, 10, 11, 12, 13, 14, 15)
                    HEAL_NMS_ROW8(16, 16, 17, 18, 19, 20, 21, 22, 23) HEAL_NMS_ROW8(24, 24, 25, 26, 27, 28, 29, 30, 31)
                } else {
                    HEAL_NMS_ROW8(0, 32, 33, 34, 35, 36, 37, 38, 39) HEAL_NMS_ROW8(8, 40, 41, 42, 43, 44, 45, 46, 47)
                    HEAL_NMS_ROW8(16, 48, 49, 50, 51, 52, 53, 54, 55) HEAL_NMS_ROW8(24, 56, 57, 58, 59, 60, 61, 62, 63)
                }
#undef HEAL_NMS_R
#undef HEAL_NMS_ROW8
            }
            const unsigned long long alive = valid & ~rem;
            if (l == 0) keep_bits[blk] = alive;
            // pick-order position of the kept rows inside gt_range
            const bool kept = (r < K) && ((alive >> l) & 1ull) && s_inr[min(r, K - 1)];
            const unsigned long long m = __ballot(kept);
            const int pos = base + __popcll(m & lanemask_lt());
            if (kept && pos < max_out) s_row[pos] = (short)r;
            base += __popcll(m);
        }
        if (l == 0) { *out_count = min(base, max_out); s_kept = min(base, max_out); }
    }
    __syncthreads();
    const int total = s_kept * 25;                                   // 24 corner floats + the score per kept box, in pick order
    for (int e0 = threadIdx.x; e0 < total; e0 += NMS_RED_THREADS * 12) {
        float v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = min(e0 + NMS_RED_THREADS * u, total - 1), pos = e / 25, k = e - pos * 25, r = s_row[pos];
            v[u] = k < 24 ? nb.corners[(size_t)r * 24 + k] : nb.scores[r];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = e0 + NMS_RED_THREADS * u, pos = e / 25, k = e - pos * 25;
            if (e < total) {
                if (k < 24) out_corners[(size_t)pos * 24 + k] = v[u];
                else out_scores[pos] = v[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_quad_iou(const float* __restrict__ a, int n,
                                                 const float* __restrict__ b, int m, float* __restrict__ iou) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * m) return;
    const int i = t / m, j = t - i * m;
    extern __shared__ __attribute__((aligned(16))) double s_poly[];   // [32][256]
    float qa[8], qb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { qa[k] = a[(size_t)i * 8 + k]; qb[k] = b[(size_t)j * 8 + k]; }
    iou[t] = quad_iou_lds<256>(qa, qb, s_poly + threadIdx.x);
}

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

struct DecWs {
    unsigned long long* cand;   // [n] candidate composites (score key << 32 | anchor index)
    uint32_t* sel;              // [top] anchor indices of the top candidates, descending
    NmsBufs nb;
};

static bool carve(Arena& a, int n, int top, DecWs& w) {
    w.cand = a.take<unsigned long long>(n);
    w.sel = a.take<uint32_t>(top);
    const int words = ceil_div(top, 64);
    w.nb.corners = a.take<float>((size_t)top * 24);
    w.nb.scores = a.take<float>(top);
    w.nb.quads = a.take<float>((size_t)top * 8);
    w.nb.inrange = a.take<int>(top);
    w.nb.mask = a.take<unsigned long long>((size_t)top * words);
    w.nb.diag = a.take<unsigned long long>((size_t)words * 64);
    w.nb.n_cand = a.take<int>(64);
    return a.ok();
}

}  // namespace heal

using namespace heal;

extern "C" size_t heal_decode_nms_workspace(int anchors_total, int nms_top) {
    Arena a(nullptr, 0);
    DecWs w;
    carve(a, anchors_total < 1 ? 1 : anchors_total, nms_top < 1 ? 1 : nms_top, w);
    return a.off + 256;
}

extern "C" int heal_decode_nms(const float* cls, const float* reg, const float* dir, const float* anchors,
                               int H, int W, int anchor_num, int num_bins, float score_thr,
                               float dir_offset, float nms_thr, int nms_top, const float* tfm_host,
                               const float* gt_range_host, float* out_corners, float* out_scores,
                               int32_t* out_count, int max_out, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(H >= 1 && W >= 1 && anchor_num >= 1, "decode_nms: bad shape");
    HEAL_REQUIRE(nms_top >= 1 && nms_top <= 1088, "decode_nms: nms_top must be in [1,1088]");
    HEAL_REQUIRE(dir == nullptr || num_bins >= 1, "decode_nms: num_bins must be >= 1");
    HEAL_REQUIRE(max_out >= 1, "decode_nms: max_out must be >= 1");
    HEAL_REQUIRE(((uintptr_t)ws & 255) == 0, "decode_nms: workspace must be 256-B aligned");
    const int n = H * W * anchor_num;
    Arena a(ws, ws_bytes);
    DecWs w;
    HEAL_REQUIRE(carve(a, n, nms_top, w), "decode_nms: workspace too small (%zu < %zu)", ws_bytes, a.off);

    DecodeParams p;
    p.H = H; p.W = W; p.A = anchor_num; p.num_bins = num_bins;
    p.score_thr = score_thr; p.dir_offset = dir_offset;
    p.period = (float)(2.0 * 3.141592653589793 / (double)(num_bins > 0 ? num_bins : 1));
    p.two_pi = (float)(2.0 * 3.141592653589793);
    for (int k = 0; k < 16; ++k) p.tfm[k] = tfm_host[k];
    for (int k = 0; k < 6; ++k) p.gt_range[k] = gt_range_host[k];
    // candidates have sigmoid score in (thr, 1]; key = bits(score) - key_base >= 1
    uint32_t key_base = 0, max_key;
    if (score_thr > 0.f) { key_base = f2u(score_thr); max_key = f2u(1.0f) - key_base; }
    else { key_base = 0; max_key = f2u(1.0f); }
    // scores are > thr so bits(score) > key_base for thr > 0; for thr <= 0 a zero score (logit -inf)
    // would collide with the "not a candidate" key 0 -- shift by one in that case
    if (score_thr <= 0.f) { key_base = 0xFFFFFFFFu; max_key += 1; }  // bits - (-1) = bits + 1
    p.key_base = key_base;
    (void)max_key;
    HEAL_REQUIRE(nms_top <= TOPK_CAP, "decode_nms: nms_top exceeds the top-k capacity");

    HEAL_FILL(w.nb.n_cand, 0, sizeof(int), s);
    k_decode_key<<<ceil_div(n, 256), 256, 0, s>>>(cls, reg, dir, anchors, p, n, w.cand, w.nb.n_cand);
    const int words = ceil_div(nms_top, 64);
    k_rank_prepare<<<TOPK_CAP / RANK_PER_BLOCK, TOPK_THREADS, 0, s>>>(cls, reg, dir, anchors, p, n, w.cand, nms_top, w.nb);
    const size_t reduce_lds = (size_t)words * 64 * words * sizeof(unsigned long long);
    HEAL_REQUIRE(reduce_lds <= 148 * 1024 && nms_top <= NMS_RED_MAXK,
                 "decode_nms: nms_top=%d needs %zu B of LDS (limit 148 KB; use <= 1088)", nms_top, reduce_lds);
    static bool attr_set = false;
    if (!attr_set) {
        HEAL_HIP(hipFuncSetAttribute((const void*)k_nms_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024));
        HEAL_HIP(hipFuncSetAttribute((const void*)k_nms_mask, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_POLY_LDS));
        attr_set = true;
    }
    k_nms_mask<<<dim3(words, words), 64 * NMS_SPLIT, NMS_POLY_LDS, s>>>(w.nb, nms_top, words, nms_thr);
    k_nms_reduce<<<1, NMS_RED_THREADS, reduce_lds, s>>>(w.nb, w.nb.diag, nms_top, words, out_corners, out_scores, out_count, max_out);
    HEAL_LAUNCH_CHECK();
    return 0;
}

extern "C" int heal_quad_iou(const float* a, int n, const float* b, int m, float* iou, void* stream) {
    if (n <= 0 || m <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        HEAL_HIP(hipFuncSetAttribute((const void*)k_quad_iou, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 256 * 8));
        attr_set = true;
    }
    k_quad_iou<<<ceil_div(n * m, 256), 256, 32 * 256 * 8, (hipStream_t)stream>>>(a, n, b, m, iou);
    HEAL_LAUNCH_CHECK();
    return 0;
}


// ---- standalone rotated NMS over quads (opencood/utils/box_utils.py:693-738 nms_rotated) ---------------------------
namespace heal {
int launch_bev_nms_walk(const unsigned long long* mask, int n, int W, unsigned long long* removed, long long* keep,
                        int* num_keep, hipStream_t s);  // iou3d.hip
}

extern "C" size_t heal_nms_quads_workspace(int n) {
    if (n < 1) n = 1;
    const size_t W = ((size_t)n + 63) / 64;
    return align_up((size_t)n * W * sizeof(unsigned long long)) + align_up(W * sizeof(unsigned long long)) + 256;
}

extern "C" int heal_nms_quads(const float* quads_sorted, int n, float thresh, void* workspace, size_t workspace_bytes,
                              long long* keep, int* num_keep, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n >= 0 && num_keep != nullptr, "nms_quads: bad arguments");
    if (n == 0) {
        HEAL_FILL(num_keep, 0, sizeof(int), s);
        return 0;
    }
    HEAL_REQUIRE(quads_sorted && keep && workspace, "nms_quads: null pointer");
    HEAL_REQUIRE(workspace_bytes >= heal_nms_quads_workspace(n) && ((uintptr_t)workspace & 7) == 0,
                 "nms_quads: workspace too small or misaligned");
    const int W = (n + 63) / 64;
    Arena a(workspace, workspace_bytes);
    NmsBufs nb;
    nb.corners = nullptr; nb.scores = nullptr; nb.inrange = nullptr; nb.n_cand = nullptr; nb.diag = nullptr;
    nb.quads = const_cast<float*>(quads_sorted);
    nb.mask = a.take<unsigned long long>((size_t)n * W);
    unsigned long long* removed = a.take<unsigned long long>(W);
    static bool mask_attr_set = false;
    if (!mask_attr_set) {
        HEAL_HIP(hipFuncSetAttribute((const void*)k_nms_mask, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NMS_POLY_LDS));
        mask_attr_set = true;
    }
    k_nms_mask<<<dim3(W, W), 64 * NMS_SPLIT, NMS_POLY_LDS, s>>>(nb, n, W, thresh);
    HEAL_LAUNCH_CHECK();
    return launch_bev_nms_walk(nb.mask, n, W, removed, keep, num_keep, s);
}
