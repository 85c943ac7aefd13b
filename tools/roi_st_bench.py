"""RoIPool forward (bf16, A alone) on large maps: the window kernels (DRN_TUNE_ROI_ST = 0) against the sparse-table kernel
(1: where it is the default, 2: forced wherever a slice fits) - bit-equality first, then time and SURVEY 8(d) bytes
(map read once + rois + the pooled matrix written once).
  python tools/roi_st_bench.py [R]"""
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
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


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


print("%-16s %-12s %10s %10s %8s" % ("map", "ROI_ST", "us", "GB/s", "of 8 TB/s"))
for (H, W, C, stride) in ((99, 151, 2048, 8), (75, 122, 2048, 8), (120, 160, 2048, 8), (150, 200, 2048, 8), (50, 76, 1024, 16), (63, 92, 1024, 16), (38, 50, 1024, 16)):
    rs = np.random.RandomState(0)
    iw, ih = W * stride, H * stride
    x0, y0 = rs.rand(R) * (iw - 40), rs.rand(R) * (ih - 40)
    rois = np.stack([np.zeros(R), x0, y0, x0 + 20 + rs.rand(R) * (iw - x0 - 20), y0 + 20 + rs.rand(R) * (ih - y0 - 20)], 1)
    rois = torch.from_numpy(rois.astype(np.float32)).to(dev)
    obj = torch.rand(R, device=dev)
    feat = (torch.randn((1, H, W, C), device=dev).relu() * 0.5).to(torch.bfloat16)
    K = C * 49
    nbytes = feat.numel() * 2 + R * K * 2 + R * 20
    outs = {}
    for st in (0, 1, 2):
        old = ops.tune(ops.TUNE_ROI_ST, st)
        try:
            A = torch.zeros((R, ops.kpad(K, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
            f = lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, out=A)
            t = timeit(f)
            outs[st] = A
        finally:
            ops.tune(ops.TUNE_ROI_ST, old)
        same = "" if st == 0 else ("  == ST 0" if torch.equal(outs[st], outs[0]) else "  DIFFERS from ST 0 (%d elements)" % int((outs[st] != outs[0]).sum()))
        print("%-16s %-12d %10.1f %10.0f %8.3f%s" % ("%dx%dx%d" % (H, W, C), st, t, nbytes / t / 1e3, nbytes / t / 1e3 / 8000.0, same), flush=True)
