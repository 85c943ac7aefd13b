"""One conv layer (or the fc7 shape) on the eight-wave kernel under its schedule variants (DRN_TUNE_PP8_VARIANT) and ring depths:
time per launch (20 launches replayed from a hipGraph) and, with PROF=1, the shader-clock split of its mainloop
(drn_tune(DRN_TUNE_PP8_PROFILE) - read + issue phase / barrier waits / MFMA phase per K slab, workgroup 0).
  python tools/pp8_probe.py [layer ...]      layers: res4_3x3 res5_3x3 res4_c1 res5_c3 c4_res4_3x3 res3_3x3 fc7"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
dev = "cuda"
LAYERS = {  # h, w, cin, cout, k, dil, residual
    "res4_3x3": (99, 151, 256, 256, 3, 2, False), "res5_3x3": (99, 151, 512, 512, 3, 2, False),
    "res4_c1": (99, 151, 1024, 256, 1, 1, False), "res5_c3": (99, 151, 512, 2048, 1, 1, True),
    "res5_c1": (99, 151, 2048, 512, 1, 1, False), "res4_c3": (99, 151, 256, 1024, 1, 1, True),
    "c4_res4_3x3": (50, 76, 256, 256, 3, 1, False), "res3_3x3": (100, 152, 128, 128, 3, 1, False),
}
names = sys.argv[1:] or ["res4_3x3", "res5_3x3", "res4_c1", "fc7"]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,5").split(",")]
STAGES = [int(v) for v in os.environ.get("STAGES", "5").split(",")]
PROF = os.environ.get("PROF", "0") == "1"


def timed(f, reps=20):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * reps) * 1e3


for name in names:
    if name == "fc7":
        M, N, K = 2000, 4096, 2048
        A = (torch.randn((M, K), device=dev) * 0.5).to(dt)
        W = (torch.randn((N, K), device=dev) * 0.03).to(dt)
        bias = torch.randn(N, device=dev) * 0.1
        out = torch.zeros((M, N), dtype=dt, device=dev)
        outT = torch.zeros((N, ops.kpad(M, dt)), dtype=dt, device=dev)
        f = lambda: ops.linear_act_fwd(A, W, M, N, K, bias, True, None, 77, 0.5, out=out, outT=outT)
        gf = 2.0 * M * N * K / 1e9
    else:
        h, w, cin, cout, k, dil, res = LAYERS[name]
        x = (torch.randn((1, h, w, cin), device=dev) * 0.5).to(dt)
        wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device=dev) * 0.05).to(dt)
        scale = torch.rand(cout, device=dev) + 0.5
        bias = torch.randn(cout, device=dev) * 0.1
        r = (torch.randn((1, h, w, cout), device=dev) * 0.5).to(dt) if res else None
        f = lambda: ops.conv2d_nhwc(x, wt, cout, k, k, 1, dil * (k // 2), dil, scale, bias, r, True)
        gf = 2.0 * h * w * k * k * cin * cout / 1e9
    ops.tune(ops.TUNE_PP8, 0)
    t = timed(f) if name != "fc7" else float("nan")
    print("%-12s %.1f GF | other kernels %.1f us" % (name, gf, t))
    ops.tune(ops.TUNE_PP8, 2)
    ops.tune(ops.TUNE_PP8_WIDE, 0)
    for st in STAGES:
        ops.tune(ops.TUNE_PP8_STAGES, st)
        for v in VARIANTS:
            if st == 3 and (v & 3):
                continue
            ops.tune(27, v)
            t = timed(f)
            print("   stages %d variant %d: %6.1f us = %5.0f TFLOP/s" % (st, v, t, gf / t * 1e3), flush=True)
            if PROF and st == 5 and (v | 8) in (8, 9, 10):
                ops.tune(27, v | 8)
                for _ in range(10):
                    f()
                torch.cuda.synchronize()
                sys.stderr.flush()
                ops.tune(28, 0)
    ops.tune(ops.TUNE_PP8_WIDE, 2)
    for v in (0, 4, 1, 5):
        ops.tune(ops.TUNE_PP8_WIDE_VARIANT, v)
        t = timed(f)
        print("   256x128 form, variant %d: %6.1f us = %5.0f TFLOP/s" % (v, t, gf / t * 1e3), flush=True)
        if PROF and v in (0, 1):
            ops.tune(ops.TUNE_PP8_WIDE_VARIANT, v | 8)
            for _ in range(10):
                f()
            torch.cuda.synchronize()
            ops.tune(28, 0)
    ops.tune(ops.TUNE_PP8_WIDE_VARIANT, 4)
    ops.tune(ops.TUNE_PP8_WIDE, 1)
    ops.tune(27, 1)
    ops.tune(ops.TUNE_PP8_STAGES, 5)
    ops.tune(ops.TUNE_PP8, 1)
