import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
def bench(nb, h, w, cin, cout, k, pin, warm, reps):
    ops.tune(ops.TUNE_CONV_RING, pin)
    x = (torch.randn((nb, h, w, cin), device="cuda") * 0.5).to(dt)
    wt = (torch.randn((cout, ops.kpad(k * k * cin, dt)), device="cuda") * 0.05).to(dt)
    scale = torch.rand(cout, device="cuda") + 0.5; bias = torch.randn(cout, device="cuda") * 0.1
    f = lambda: ops.conv2d_nhwc(x, wt, cout, k, k, 1, k // 2, 1, scale, bias, None, True)
    for _ in range(3): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    for _ in range(warm): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (20 * reps) * 1e3
for warm, reps in ((1, 5), (50, 50), (500, 200)):
    for pin in (64, 0):
        print("warm %3d reps %3d pin %3d  res4 3x3 256 Nb=1: %.1f us" % (warm, reps, pin, bench(1, 50, 76, 256, 256, 3, pin, warm, reps)))
# eager back-to-back launches (no graph)
ops.tune(ops.TUNE_CONV_RING, 64)
x = (torch.randn((1, 50, 76, 256), device="cuda") * 0.5).to(dt)
wt = (torch.randn((256, ops.kpad(9 * 256, dt)), device="cuda") * 0.05).to(dt)
scale = torch.rand(256, device="cuda") + 0.5; bias = torch.randn(256, device="cuda") * 0.1
y = torch.empty((1, 50, 76, 256), device="cuda", dtype=dt)
f = lambda: ops.conv2d_nhwc(x, wt, 256, 3, 3, 1, 1, 1, scale, bias, None, True)
for _ in range(200): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(1000): f()
b.record(); torch.cuda.synchronize()
print("eager 1000 launches: %.1f us each" % (a.elapsed_time(b)))
