"""The training pooling launch (roi_pool7_map64_kernel: 14x14x1024 bf16 map, 2000 SURVEY 8(d) boxes -> A and A^T)
replayed from a hipGraph (the Python call costs ~50 us - more than the kernel's phases - so eager timing is host-bound),
optionally under tune knobs: python tools/roi_pool_probe.py [knob=value ...]   (10 = chunks per workgroup, 11 = prefetch)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
for kv in sys.argv[1:]:
    ops.tune(*[int(x) for x in kv.split("=")])
dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (R, C, H, W, stride) in ((2000, 1024, 14, 14, 16), (4000, 2048, 27, 27, 8)):
    g = torch.Generator().manual_seed(7)
    feat = torch.rand((1, H, W, C), generator=g).to(dev).to(torch.bfloat16)
    S = 224
    x0, y0 = torch.rand(R, generator=g) * (S - 40), torch.rand(R, generator=g) * (S - 40)
    bw, bh = 20 + torch.rand(R, generator=g) * (S - x0 - 20), 20 + torch.rand(R, generator=g) * (S - y0 - 20)
    rois = torch.stack([torch.zeros(R), x0, y0, (x0 + bw).clamp(max=S), (y0 + bh).clamp(max=S)], 1).to(dev)
    obj = torch.sort(torch.rand(R, generator=g), descending=True).values.to(dev)
    K1 = C * 49
    A = torch.zeros((R, ops.kpad(K1, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
    AT = torch.zeros((K1, ops.kpad(R, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
    t = timed(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, out=A, out_t=AT))
    nb = 2 * R * K1 * 2
    print("%s  R=%d C=%d %dx%d  A+A^T %.1f MB  %.1f us  %.2f TB/s" % (" ".join(sys.argv[1:]) or "default", R, C, H, W,
                                                                 nb / 1e6, t, nb / t / 1e6))
    # A alone (what the pooling launch would cost if the fc6 dW read A itself): the 8-ROI whole-map kernel, then the
    # 64-ROI kernel without its A^T store loop
    for knob, label in ((0, "A only, 8-ROI kernel"), (1, "A only, 64-ROI kernel")):
        ops.tune(ops.TUNE_ROI_MAP64_A, knob)
        t = timed(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / stride, out=A))
        print("    %-24s %.1f MB  %.1f us  %.2f TB/s" % (label, nb / 2e6, t, nb / 2 / t / 1e6))
    ops.tune(ops.TUNE_ROI_MAP64_A, 0)
