"""A few eager launches of chosen trunk layers for `rocprofv3 --kernel-trace --stats` (true kernel durations, without the launch
gaps a back-to-back timing includes).  usage: python tools/conv_trace.py [ring-pin]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
if len(sys.argv) > 1:
    ops.tune(ops.TUNE_CONV_RING, int(sys.argv[1]))
L = [(50, 76, 512, 256, 1, False), (50, 76, 1024, 256, 1, False), (50, 76, 256, 256, 3, False), (50, 76, 256, 1024, 1, True),
     (100, 152, 128, 128, 3, False), (100, 152, 128, 512, 1, True), (100, 152, 512, 128, 1, False)]
for h, w, cin, cout, k, res in L:
    x = (torch.randn((1, h, w, cin), device="cuda") * 0.5).to(dt)
    wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device="cuda") * 0.05).to(dt)
    scale = torch.rand(cout, device="cuda") + 0.5; bias = torch.randn(cout, device="cuda") * 0.1
    r = (torch.randn((1, h, w, cout), device="cuda") * 0.5).to(dt) if res else None
    for _ in range(30):
        ops.conv2d_nhwc(x, wt, cout, k, k, 1, k // 2, 1, scale, bias, r, True)
    torch.cuda.synchronize()
