"""ROIAlign (detectron2/layers/roi_align.py:22-117; csrc/ROIAlign/ROIAlign_cuda.cu:65-250) forward and backward timing on the
maps the trunks produce - 14x14x1024 (the bench shape), 50x76x1024 (C4 at 800x1216), 99x151x2048 (dilated C5) - R = 2000,
sampling_ratio 0 (adaptive) and 2, aligned = True (POOLER_TYPE ROIAlignV2), bf16 and fp32, against SURVEY 8(d)'s bytes
(map read once + rois + the pooled matrix written once; backward: the pooled gradient read once + the fp32 map gradient
written once), with RoIPool on the same inputs beside it.
  python tools/roi_align_bench.py"""
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
R = 2000


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("%-22s %-6s %-22s %10s %10s %8s" % ("map", "dtype", "op", "us", "GB/s", "of 8 TB/s"))
for (H, W, C, stride) in ((14, 14, 1024, 16), (50, 76, 1024, 16), (99, 151, 2048, 8)):
    rs = np.random.RandomState(0)
    iw, ih = W * stride, H * stride
    x0, y0 = rs.rand(R) * (iw - 40), rs.rand(R) * (ih - 40)
    rois = np.stack([np.zeros(R), x0, y0, x0 + 20 + rs.rand(R) * (iw - x0 - 20), y0 + 20 + rs.rand(R) * (ih - y0 - 20)], 1)
    rois = torch.from_numpy(rois.astype(np.float32)).to(dev)
    obj = torch.rand(R, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        es = 2 if dt == torch.bfloat16 else 4
        feat = (torch.randn((1, H, W, C), device=dev).relu() * 0.5).to(dt)
        K = C * 49
        A = torch.zeros((R, ops.kpad(K, dt)), dtype=dt, device=dev)
        fwd_bytes = feat.numel() * es + R * K * es + R * 20
        cases = [("RoIPool fwd", lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, out=A), fwd_bytes)]
        for sr in (0, 2):
            cases.append(("ROIAlign fwd sr=%d" % sr,
                          lambda sr=sr: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, mode=1, sampling_ratio=sr, aligned=True, out=A), fwd_bytes))
        if C * H * W <= 1024 * 50 * 76:  # (the backward's fp32 map gradient: the sizes a trainable trunk would see)
            g = (torch.randn((R, ops.kpad(K, dt)), device=dev) * 0.1).to(dt)
            bwd_bytes = R * K * es + H * W * C * 4 + R * 20
            for sr in (0, 2):
                cases.append(("ROIAlign bwd sr=%d" % sr,
                              lambda sr=sr: ops.roi_pool_backward_nhwc(g, rois, obj, (1, H, W, C), 7, 1.0 / stride, mode=1, sampling_ratio=sr,
                                                                        aligned=True), bwd_bytes))
        for name, f, nbytes in cases:
            t = timeit(f)
            print("%-22s %-6s %-22s %10.1f %10.0f %8.3f" % ("%dx%dx%d" % (H, W, C), "bf16" if es == 2 else "fp32", name, t, nbytes / t / 1e3,
                                                            nbytes / t / 1e3 / 8000.0), flush=True)
