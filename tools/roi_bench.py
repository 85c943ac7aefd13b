"""Time the ROIPool launch of BASELINE configs[1] (14x14x1024 bf16 map, 2000 proposals) with and without the fused
transposed output, plus the stand-alone transpose for comparison."""
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
H = W = int(os.environ.get("HW", 14))
C, R = 1024, 2000
feat = torch.randn((1, H, W, C), device=dev).to(torch.bfloat16)
rs = np.random.RandomState(0)
x0, y0 = rs.rand(R) * 150, rs.rand(R) * 150
rois = np.stack([np.zeros(R), x0, y0, x0 + 20 + rs.rand(R) * (204 - x0), y0 + 20 + rs.rand(R) * (204 - y0)], 1)
rois = torch.from_numpy(rois.astype(np.float32)).to(dev)
obj = torch.rand(R, device=dev)
K = C * 49
A = torch.zeros((R, K), dtype=torch.bfloat16, device=dev)
AT = torch.zeros((K, ops.kpad(R, torch.bfloat16)), dtype=torch.bfloat16, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("pool A only      %7.1f us" % timeit(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A)))
print("pool A + A^T     %7.1f us" % timeit(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A, out_t=AT)))
print("transpose alone  %7.1f us" % timeit(lambda: ops.transpose2d(A, R, K, out=AT)))
for on in (0, 256, 512, 1024):
    ops.tune(ops.TUNE_ROI_MAP64, on)
    t = timeit(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A, out_t=AT))
    nbytes = 2 * R * K * 2 + feat.numel() * 2
    print("map64=%d  pool A + A^T  %7.1f us  (%.2f TB/s of %.0f MB)" % (on, t, nbytes / t / 1e6, nbytes / 1e6))
ops.tune(ops.TUNE_ROI_MAP64, 512)
