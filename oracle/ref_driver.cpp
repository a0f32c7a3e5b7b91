// Driver around the REFERENCE's own CPU IoU (det3d/ops/iou3d_nms/src/iou3d_cpu.cpp, compiled
// unmodified from /root/reference by oracle/build_ref.py into oracle/_ref/).  It only adapts the
// reference's at::Tensor entry points (iou3d_cpu.h: boxes_iou_bev_cpu, boxes_aligned_iou_bev_cpu)
// to plain pointers so that Python can call them through ctypes.  Test infrastructure only.
#include <torch/torch.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);
int boxes_aligned_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int ref_boxes_iou_bev_cpu(const float* a, int na, const float* b, int nb, float* out) {
  auto opt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor ta = torch::from_blob(const_cast<float*>(a), {na, 7}, opt);
  at::Tensor tb = torch::from_blob(const_cast<float*>(b), {nb, 7}, opt);
  at::Tensor to = torch::from_blob(out, {na, nb}, opt);
  return boxes_iou_bev_cpu(ta, tb, to);
}

extern "C" int ref_boxes_aligned_iou_bev_cpu(const float* a, const float* b, int n, float* out) {
  auto opt = torch::TensorOptions().dtype(torch::kFloat32);
  at::Tensor ta = torch::from_blob(const_cast<float*>(a), {n, 7}, opt);
  at::Tensor tb = torch::from_blob(const_cast<float*>(b), {n, 7}, opt);
  at::Tensor to = torch::from_blob(out, {n, 1}, opt);
  return boxes_aligned_iou_bev_cpu(ta, tb, to);
}
