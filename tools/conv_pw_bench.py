"""The tail of a res2 bottleneck at a real image size: 3x3 (64 -> 64) + 1x1 (64 -> 256, + shortcut, ReLU) as two launches
against drn_conv3x3_pw_nhwc (one launch, the 3x3's output never in memory); each replayed 20x from a hipGraph.
  python tools/conv_pw_bench.py [H W]     (image size; default 800 1216 -> a 200 x 304 map)"""
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1216)
h, w = H // 4, W // 4
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
x = (torch.randn((1, h, w, 64), device=dev) * 0.5).to(dt)
res = (torch.randn((1, h, w, 256), device=dev) * 0.5).to(dt)


def pack(cout, k):
    wt = torch.randn((cout, k), device=dev) * math.sqrt(2.0 / k)
    ld = (k * 2 + 127) // 128 * 64
    out = torch.zeros((cout, ld), dtype=dt, device=dev)
    out[:, :k] = wt.to(dt)
    return out


w2, w3 = pack(64, 576), pack(256, 64)
s2, b2 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
s3, b3 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
y2 = torch.empty((1, h, w, 64), dtype=dt, device=dev)


def two():
    y = ops.conv2d_nhwc(x, w2, 64, 3, 3, 1, 1, 1, s2, b2, None, True)
    return ops.conv2d_nhwc(y, w3, 256, 1, 1, 1, 0, 1, s3, b3, res, True)


def one():
    return ops.conv3x3_pw_nhwc(x, w2, s2, b2, True, w3, s3, b3, res, 1.0, True)


assert torch.equal(two(), one())


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3


t2, t1 = timeit(two), timeit(one)
gf = 2.0 * h * w * (576 * 64 + 64 * 256) / 1e9
print("res2 tail @ %dx%d (%d pixels): 3x3 + 1x1 as two launches %.1f us, fused %.1f us (%.0f TFLOP/s); bit-identical"
      % (H, W, h * w, t2, t1, gf / t1 / 1e-3))
