"""Per-layer timing of the trunk's conv kernel on the WS-ResNet50 C4 layer shapes (FrozenBN affine + ReLU / residual
epilogue as in the model), stand-alone: each layer is replayed 20x from a hipGraph so launch gaps do not count.
Reports us, TFLOP/s and the algorithmic HBM rate (input + weights + output + residual, once each).
  python tools/conv_bench.py [H W]      (image size; default 800 1216)
CONV_DTYPE=fp8: the fp8 trunk's layers (drn_conv2d_nhwc_q: fp8 x / w / y / residual), TFLOP/s against the 5 PFLOP/s dense
fp8 peak; CONV_FP8_K64=0 runs them on the K = 16 non-scaled MFMA (round 2's path) for the A/B."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1216)
FP8 = os.environ.get("CONV_DTYPE", "bf16") == "fp8"
dt = torch.float8_e4m3fn if FP8 else torch.bfloat16
es = 1 if FP8 else 2
if FP8:
    ops.tune(ops.TUNE_FP8_K64, int(os.environ.get("CONV_FP8_K64", "1")))
dev = "cuda"
# (name, stride-of-map, cin, cout, k, residual, count per trunk)
LAYERS = [("stem.conv2/3 3x3 64", 2, 64, 64, 3, False, 2),
          ("res2 conv1 1x1 64>64", 4, 64, 64, 1, False, 1), ("res2 conv1 1x1 256>64", 4, 256, 64, 1, False, 2),
          ("res2 conv2 3x3 64", 4, 64, 64, 3, False, 3), ("res2 conv3 1x1 64>256 +res", 4, 64, 256, 1, True, 3),
          ("res3 conv1 1x1 256>128", 8, 256, 128, 1, False, 1), ("res3 conv1 1x1 512>128", 8, 512, 128, 1, False, 3),
          ("res3 conv2 3x3 128", 8, 128, 128, 3, False, 4), ("res3 conv3 1x1 128>512 +res", 8, 128, 512, 1, True, 4),
          ("res4 conv1 1x1 512>256", 16, 512, 256, 1, False, 1), ("res4 conv1 1x1 1024>256", 16, 1024, 256, 1, False, 5),
          ("res4 conv2 3x3 256", 16, 256, 256, 3, False, 6), ("res4 conv3 1x1 256>1024 +res", 16, 256, 1024, 1, True, 6)]
if os.environ.get("CONV_BENCH_EXTRA"):  # the +res layers without their residual: what the epilogue's reads cost
    LAYERS += [("res2 conv3 1x1 64>256 (no res)", 4, 64, 256, 1, False, 0), ("res3 conv3 1x1 128>512 (no res)", 8, 128, 512, 1, False, 0),
               ("res4 conv3 1x1 256>1024 (no res)", 16, 256, 1024, 1, False, 0)]
tot_us = tot_gf = 0.0
print("%-30s %9s %8s %8s %9s %8s" % ("layer @ %dx%d" % (H, W), "pixels", "us", "TFLOP/s", "GB/s", "x count"))
for name, s, cin, cout, k, res, cnt in LAYERS:
    h, w = H // s, W // s
    x = (torch.randn((1, h, w, cin), device=dev) * 0.5).to(dt)
    wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device=dev) * 0.05).to(dt)
    scale = torch.rand(cout, device=dev) + 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    r = (torch.randn((1, h, w, cout), device=dev) * 0.5).to(dt) if res else None
    if FP8:
        f = lambda: ops.conv2d_nhwc_q(x, wt, cout, k, k, 1, k // 2, 1, scale, bias, dt, r, 1.0, True)
    else:
        f = lambda: ops.conv2d_nhwc(x, wt, cout, k, k, 1, k // 2, 1, scale, bias, r, True)
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            y = f()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 100 * 1e3
    gf = 2.0 * h * w * k * k * cin * cout / 1e9
    mb = (h * w * (cin + cout * (2 if res else 1)) + k * k * cin * cout) * es / 1e6
    print("%-30s %9d %8.1f %8.1f %9.0f %8d" % (name, h * w, us, gf / us * 1e3, mb / us * 1e3, cnt))
    tot_us += us * cnt
    tot_gf += gf * cnt
print("sum over the trunk's convs (stem.conv1 excluded): %.0f us, %.1f GF -> %.0f TFLOP/s = %.3f of the dense %s MFMA peak" % (
    tot_us, tot_gf, tot_gf / tot_us * 1e3, tot_gf / tot_us * 1e3 / (5000.0 if FP8 else 2500.0), "fp8" if FP8 else "bf16"))
