// Binding-only translation unit for oracle/_ref: exposes the reference's own CPU rotated-IoU routine
// (/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232, compiled from where it lies) to Python.
// No reference source is copied; this file only declares the symbol and registers it.
#include <torch/extension.h>
int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu, "reference CPU rotated BEV IoU (N,7)x(M,7)->(N,M)");
}
