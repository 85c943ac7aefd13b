"""Eager launches of the shipped DC5 recipe's large layers at 800x1216 (stride-8 map 100x152 = 15200 pixels) for PMC / trace passes:
res5 1x1 (2048 -> 512, 512 -> 2048 + shortcut, 1024 -> 2048: conv1x1_pp_kernel), res5 3x3 dil 2 (conv_nhwc_kernel<128,128>),
res4 3x3 dil 2 and res4 1x1s (conv_ring_kernel / conv_nhwc_kernel)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
H, W = 100, 152
L = [(2048, 512, 1, 1, False), (512, 2048, 1, 1, True), (1024, 2048, 1, 1, False), (512, 512, 3, 2, False),
     (256, 256, 3, 2, False), (1024, 256, 1, 1, False), (256, 1024, 1, 1, True)]
for cin, cout, k, dil, res in L:
    x = (torch.randn((1, H, W, cin), device="cuda") * 0.5).to(dt)
    wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device="cuda") * 0.05).to(dt)
    scale = torch.rand(cout, device="cuda") + 0.5; bias = torch.randn(cout, device="cuda") * 0.1
    r = (torch.randn((1, H, W, cout), device="cuda") * 0.5).to(dt) if res else None
    for _ in range(8):
        ops.conv2d_nhwc(x, wt, cout, k, k, 1, dil * (k // 2), dil, scale, bias, r, True)
    torch.cuda.synchronize()
print("done")
