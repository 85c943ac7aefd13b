// ORACLE (test infrastructure). Minimal pybind shim of OURS around the reference's
// ROIAlign_forward_cpu / ROIAlign_backward_cpu, which are compiled from
// /root/reference/detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp where it lies (see
// oracle/build.py). Declarations follow detectron2/layers/csrc/ROIAlign/ROIAlign.h:9-30.
#include <torch/extension.h>

namespace detectron2 {
at::Tensor ROIAlign_forward_cpu(const at::Tensor& input, const at::Tensor& rois,
                                const float spatial_scale, const int pooled_height,
                                const int pooled_width, const int sampling_ratio, bool aligned);
at::Tensor ROIAlign_backward_cpu(const at::Tensor& grad, const at::Tensor& rois,
                                 const float spatial_scale, const int pooled_height,
                                 const int pooled_width, const int batch_size, const int channels,
                                 const int height, const int width, const int sampling_ratio,
                                 bool aligned);
}  // namespace detectron2

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("roi_align_forward", &detectron2::ROIAlign_forward_cpu);
  m.def("roi_align_backward", &detectron2::ROIAlign_backward_cpu);
}
