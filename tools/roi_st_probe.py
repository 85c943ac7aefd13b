"""Phase profile of the sparse-table RoIPool kernel (profile build: drn_tune(31, 10); counters printed by drn_tune(31, 12)):
shader-clock cycles per block for the class scan, the slice staging, the row / column doubling steps and the pooling of the
listed ROIs, as thread 0 sees them (barrier waits included).
  python tools/roi_st_probe.py"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"
for (H, W, C, stride, st) in ((99, 151, 2048, 8, 2), (75, 122, 2048, 8, 2)):
    for R in (250, 2000):
        rs = np.random.RandomState(0)
        iw, ih = W * stride, H * stride
        x0, y0 = rs.rand(R) * (iw - 40), rs.rand(R) * (ih - 40)
        rois = np.stack([np.zeros(R), x0, y0, x0 + 20 + rs.rand(R) * (iw - x0 - 20), y0 + 20 + rs.rand(R) * (ih - y0 - 20)], 1)
        rois = torch.from_numpy(rois.astype(np.float32)).to(dev)
        obj = torch.rand(R, device=dev)
        feat = (torch.randn((1, H, W, C), device=dev).relu() * 0.5).to(torch.bfloat16)
        A = torch.zeros((R, ops.kpad(C * 49, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
        old = ops.tune(ops.TUNE_ROI_ST, st)
        ops.tune(ops.TUNE_ROI_ST, 10)
        try:
            for _ in range(4):
                ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, out=A)
            torch.cuda.synchronize()
            sys.stderr.write("%dx%dx%d R=%d: " % (H, W, C, R))
            sys.stderr.flush()
            ops.tune(ops.TUNE_ROI_ST, 12)
        finally:
            ops.tune(ops.TUNE_ROI_ST, 11)
            ops.tune(ops.TUNE_ROI_ST, old)
