// ORACLE (test infrastructure). Minimal pybind shim of OURS around the reference's pcl_loss_forward /
// pcl_loss_backward (the inline dispatchers of projects/WSL/wsl/layers/csrc/pcl_loss/pcl_loss.h:52-131, which always
// take the CPU branch), compiled together with projects/WSL/wsl/layers/csrc/pcl_loss/pcl_loss_cpu.cpp where it lies
// (see oracle/build.py).  Mirrors the two m.def lines of projects/WSL/wsl/layers/csrc/vision.cpp for this op.
#include <torch/extension.h>

#include "pcl_loss/pcl_loss.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("pcl_loss_forward", &wsl::pcl_loss_forward);
  m.def("pcl_loss_backward", &wsl::pcl_loss_backward);
}
