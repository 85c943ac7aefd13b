"""Per-layer timing of the trunk's conv kernels on the WS-ResNet50 layer shapes (FrozenBN affine + ReLU / residual epilogue as
in the model), stand-alone: each layer is replayed 20x from a hipGraph so launch gaps do not count.
Reports us, TFLOP/s and the algorithmic HBM rate (input + weights + output + residual, once each).
  python tools/conv_bench.py [H W] [--workload r50c4|r50dc5]      (image size; default 800 1216)
r50c4: the constructed C4 trunk of BASELINE configs[1] (stem, res2, res3, res4 at strides 4 / 8 / 16).
r50dc5: the SHIPPED recipe (projects/WSL/configs/PascalVOC-Detection/oicr_WSR_50_DC5_1x.yaml; resnet_ws.py:616-703 with
RES5_DILATION = 2): res2 pooled to stride 8, res3's pool has stride 1, res4 AND res5 run dilated (2) at stride 8.
CONV_RING=0|64|128 pins drn_tune(DRN_TUNE_CONV_RING) (0 = the register-staged kernels of rounds 1-4).
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
argv = [a for a in sys.argv[1:]]
WORKLOAD = "r50c4"
if "--workload" in argv:
    i = argv.index("--workload")
    WORKLOAD = argv[i + 1]
    del argv[i:i + 2]
H, W = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (800, 1216)
FP8 = os.environ.get("CONV_DTYPE", "bf16") == "fp8"
dt = torch.float8_e4m3fn if FP8 else torch.bfloat16
es = 1 if FP8 else 2
if FP8:
    ops.tune(ops.TUNE_FP8_K64, int(os.environ.get("CONV_FP8_K64", "1")))
if "CONV_RING" in os.environ:
    ops.tune(ops.TUNE_CONV_RING, int(os.environ["CONV_RING"]))
dev = "cuda"
h2, w2 = H // 2, W // 2            # stem.conv1 output (stride-2 conv)
h4, w4 = h2 // 2, w2 // 2          # after the stem's pool: res2
h8, w8 = h4 // 2, w4 // 2          # after res2's pool: res3
# (name, h, w, cin, cout, k, dil, residual, count per trunk)
LAYERS = [("stem.conv2/3 3x3 64", h2, w2, 64, 64, 3, 1, False, 2),
          ("res2 conv1 1x1 64>64", h4, w4, 64, 64, 1, 1, False, 1), ("res2 conv1 1x1 256>64", h4, w4, 256, 64, 1, 1, False, 2),
          ("res2 conv2 3x3 64", h4, w4, 64, 64, 3, 1, False, 3), ("res2 conv3 1x1 64>256 +res", h4, w4, 64, 256, 1, 1, True, 3),
          ("res2 shortcut 1x1 64>256", h4, w4, 64, 256, 1, 1, False, 1),
          ("res3 conv1 1x1 256>128", h8, w8, 256, 128, 1, 1, False, 1), ("res3 conv1 1x1 512>128", h8, w8, 512, 128, 1, 1, False, 3),
          ("res3 conv2 3x3 128", h8, w8, 128, 128, 3, 1, False, 4), ("res3 conv3 1x1 128>512 +res", h8, w8, 128, 512, 1, 1, True, 4),
          ("res3 shortcut 1x1 256>512", h8, w8, 256, 512, 1, 1, False, 1)]
if WORKLOAD == "r50c4":
    h16, w16 = h8 // 2, w8 // 2
    LAYERS += [("res4 conv1 1x1 512>256", h16, w16, 512, 256, 1, 1, False, 1), ("res4 conv1 1x1 1024>256", h16, w16, 1024, 256, 1, 1, False, 5),
               ("res4 conv2 3x3 256", h16, w16, 256, 256, 3, 1, False, 6), ("res4 conv3 1x1 256>1024 +res", h16, w16, 256, 1024, 1, 1, True, 6),
               ("res4 shortcut 1x1 512>1024", h16, w16, 512, 1024, 1, 1, False, 1)]
elif WORKLOAD == "r50dc5":
    hd, wd = h8 - 1, w8 - 1  # res3's 2x2 pool with stride 1
    LAYERS += [("res4 conv1 1x1 512>256", hd, wd, 512, 256, 1, 1, False, 1), ("res4 conv1 1x1 1024>256", hd, wd, 1024, 256, 1, 1, False, 5),
               ("res4 conv2 3x3 256 dil 2", hd, wd, 256, 256, 3, 2, False, 6), ("res4 conv3 1x1 256>1024 +res", hd, wd, 256, 1024, 1, 1, True, 6),
               ("res4 shortcut 1x1 512>1024", hd, wd, 512, 1024, 1, 1, False, 1),
               ("res5 conv1 1x1 1024>512", hd, wd, 1024, 512, 1, 1, False, 1), ("res5 conv1 1x1 2048>512", hd, wd, 2048, 512, 1, 1, False, 2),
               ("res5 conv2 3x3 512 dil 2", hd, wd, 512, 512, 3, 2, False, 3), ("res5 conv3 1x1 512>2048 +res", hd, wd, 512, 2048, 1, 1, True, 3),
               ("res5 shortcut 1x1 1024>2048", hd, wd, 1024, 2048, 1, 1, False, 1)]
else:
    raise SystemExit("unknown workload " + WORKLOAD)
if os.environ.get("CONV_BENCH_EXTRA"):  # the +res layers without their residual: what the epilogue's reads cost
    LAYERS += [("res2 conv3 1x1 64>256 (no res)", h4, w4, 64, 256, 1, 1, False, 0), ("res3 conv3 1x1 128>512 (no res)", h8, w8, 128, 512, 1, 1, False, 0)]
tot_us = tot_gf = 0.0
print("%-30s %9s %8s %8s %9s %8s" % ("%s @ %dx%d" % (WORKLOAD, H, W), "pixels", "us", "TFLOP/s", "GB/s", "x count"))
ONLY = os.environ.get("CONV_ONLY")  # substring filter on the layer names
for name, h, w, cin, cout, k, dil, res, cnt in LAYERS:
    if ONLY and ONLY not in name:
        continue
    x = (torch.randn((1, h, w, cin), device=dev) * 0.5).to(dt)
    wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device=dev) * 0.05).to(dt)
    scale = torch.rand(cout, device=dev) + 0.5
    bias = torch.randn(cout, device=dev) * 0.1
    r = (torch.randn((1, h, w, cout), device=dev) * 0.5).to(dt) if res else None
    pad = dil * (k // 2)
    if FP8:
        f = lambda: ops.conv2d_nhwc_q(x, wt, cout, k, k, 1, pad, dil, scale, bias, dt, r, 1.0, True)
    else:
        f = lambda: ops.conv2d_nhwc(x, wt, cout, k, k, 1, pad, dil, scale, bias, r, True)
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
