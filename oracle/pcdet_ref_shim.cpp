// TEST INFRASTRUCTURE (oracle/): C-ABI glue around the reference's OWN CPU rotated-IoU routine.
//
// `boxes_iou_bev_cpu` is defined in /root/reference/opencood/pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp:233-252 and is
// compiled from that file where it lies (oracle/Makefile.ref); nothing of the reference is copied here.  This shim
// only wraps caller memory as CPU tensors so the routine can be called through ctypes.
#include <torch/extension.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int ref_boxes_iou_bev_cpu(const float* boxes_a, int n, const float* boxes_b, int m, float* iou_out) {
    auto opt = torch::TensorOptions().dtype(torch::kFloat32);
    at::Tensor a = torch::from_blob(const_cast<float*>(boxes_a), {n, 7}, opt);
    at::Tensor b = torch::from_blob(const_cast<float*>(boxes_b), {m, 7}, opt);
    at::Tensor o = torch::from_blob(iou_out, {n, m}, opt);
    return boxes_iou_bev_cpu(a, b, o);
}
