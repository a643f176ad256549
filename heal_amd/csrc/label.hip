// Training-side anchor labelling: the IoU / assignment core of VoxelPostprocessor.generate_label
// (opencood/data_utils/post_processor/voxel_postprocessor.py:139-165) on the device.
//
//   k_label_iou     one thread per anchor: axis-aligned ("stand-up") IoU against every ground-truth box with the
//                   arithmetic of the reference's Cython bbox_overlaps (opencood/utils/box_overlaps.pyx:17-57: float32
//                   differences, `+ 1` in double, one rounding per stored float32) -> first gt above pos_threshold,
//                   "all below neg_threshold" flag, and a per-gt arg-max over anchors (first maximum) through a packed
//                   64-bit atomicMax (IoU bits high, ~anchor low)
//   k_label_best    one thread: the best anchor of every gt (IoU > 0) becomes positive if it is not already (smallest gt
//                   index wins, as np.unique keeps the first occurrence) and is cleared from the negatives
#include "common.h"
#include "../../include/heal_amd.h"

namespace heal {

__device__ __forceinline__ float standup_iou(const float4 b, const float4 q) {
    // box_overlaps.pyx:36-56 with the types Cython generates
    const float box_area = (float)(((double)(q.z - q.x) + 1.0) * ((double)(q.w - q.y) + 1.0));
    const float iw = (float)((double)(fminf(b.z, q.z) - fmaxf(b.x, q.x)) + 1.0);
    if (!(iw > 0.f)) return 0.f;
    const float ih = (float)((double)(fminf(b.w, q.w) - fmaxf(b.y, q.y)) + 1.0);
    if (!(ih > 0.f)) return 0.f;
    const float inter = iw * ih;
    const float ua = (float)((((double)(b.z - b.x) + 1.0) * ((double)(b.w - b.y) + 1.0) + (double)box_area) -
                             (double)inter);
    return inter / ua;
}

constexpr int LABEL_MAX_GT = 512;

__global__ __launch_bounds__(256) void k_label_iou(const float4* __restrict__ anchors, int n_anchors,
                                                  const float4* __restrict__ gts, int n_gt, float pos_thr,
                                                  float neg_thr, int* __restrict__ assigned,
                                                  unsigned char* __restrict__ neg,
                                                  unsigned long long* __restrict__ best /*[n_gt]*/) {
    __shared__ float4 sgt[LABEL_MAX_GT];
    __shared__ unsigned long long sbest[LABEL_MAX_GT];
    for (int g = threadIdx.x; g < n_gt; g += 256) { sgt[g] = gts[g]; sbest[g] = 0ull; }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < n_anchors) {
        const float4 b = anchors[a];
        int first = -1;
        bool all_below = true;
        for (int g = 0; g < n_gt; ++g) {
            const float v = standup_iou(b, sgt[g]);
            if (first < 0 && v > pos_thr) first = g;
            if (!(v < neg_thr)) all_below = false;
            // arg-max over anchors, first maximum: larger IoU wins, then the smaller anchor index
            const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(~(unsigned)a);
            if (key > sbest[g]) atomicMax(&sbest[g], key);
        }
        assigned[a] = first;
        neg[a] = all_below ? 1 : 0;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < n_gt; g += 256)
        if (sbest[g]) atomicMax(&best[g], sbest[g]);
}

__global__ void k_label_best(const unsigned long long* __restrict__ best, int n_gt, int* __restrict__ assigned,
                             unsigned char* __restrict__ neg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int g = 0; g < n_gt; ++g) {
        const unsigned long long key = best[g];
        const float v = __uint_as_float((unsigned)(key >> 32));
        if (!(v > 0.f)) continue;  // "make sure all highest iou is larger than 0" (voxel_postprocessor.py:151-153)
        const int a = (int)(~(unsigned)(key & 0xFFFFFFFFull));
        if (assigned[a] < 0) assigned[a] = g;
        neg[a] = 0;
    }
}

}  // namespace heal

using namespace heal;

extern "C" size_t heal_label_assign_workspace(int n_gt) {
    return align_up((size_t)(n_gt < 1 ? 1 : n_gt) * sizeof(unsigned long long)) + 256;
}

extern "C" int heal_label_assign(const float* anchor_boxes, int n_anchors, const float* gt_boxes, int n_gt,
                                 float pos_threshold, float neg_threshold, int32_t* assigned, uint8_t* neg,
                                 void* ws, size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HEAL_REQUIRE(n_anchors >= 0 && n_gt >= 0, "label_assign: negative count");
    HEAL_REQUIRE(n_gt <= LABEL_MAX_GT, "label_assign: at most %d ground-truth boxes (got %d)", LABEL_MAX_GT, n_gt);
    if (n_anchors == 0) return 0;
    HEAL_REQUIRE(anchor_boxes && assigned && neg && (n_gt == 0 || gt_boxes), "label_assign: null pointer");
    HEAL_REQUIRE(ws_bytes >= heal_label_assign_workspace(n_gt) && ((uintptr_t)ws & 7) == 0,
                 "label_assign: workspace too small or misaligned");
    unsigned long long* best = (unsigned long long*)ws;
    HEAL_FILL(best, 0, sizeof(unsigned long long) * (size_t)(n_gt < 1 ? 1 : n_gt), s);
    k_label_iou<<<ceil_div(n_anchors, 256), 256, 0, s>>>(reinterpret_cast<const float4*>(anchor_boxes), n_anchors,
                                                        reinterpret_cast<const float4*>(gt_boxes), n_gt, pos_threshold,
                                                        neg_threshold, assigned, neg, best);
    if (n_gt > 0) k_label_best<<<1, 64, 0, s>>>(best, n_gt, assigned, neg);
    HEAL_LAUNCH_CHECK();
    return 0;
}
