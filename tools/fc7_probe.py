import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
M, N, K = 2000, 2048, 2048
def timeit(f, n=30):
    for _ in range(5): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
x = (torch.randn((1, 40, 50, K), device="cuda") * 0.5).to(dt)       # 2000 "pixels"
wt = (torch.randn((N, K), device="cuda") * 0.05).to(dt)
scale = torch.ones(N, device="cuda"); bias = torch.randn(N, device="cuda") * 0.1
for pin in (0, 64, 128, 1):
    ops.tune(ops.TUNE_CONV_RING, pin)
    for pp in (0, 2):
        ops.tune(ops.TUNE_CONV_PP, pp if pp else 1)
        t = timeit(lambda: ops.conv2d_nhwc(x, wt, N, 1, 1, 1, 0, 1, scale, bias, None, True))
        print("conv1x1 as GEMM [2000 x 2048 x 2048] + bias + ReLU: ring pin %3d, pp knob %d: %.1f us = %.0f TFLOP/s" % (pin, pp if pp else 1, t, 2.0 * M * N * K / t * 1e-6))
A = x.view(M, K)
for s in (1, 2, 4):
    out = torch.empty((s, M, N), dtype=torch.float32, device="cuda")
    t = timeit(lambda: ops.gemm_nt(A, wt, M, N, K, out=out, splits=s))
    print("gemm_nt splits=%d: %.1f us = %.0f TFLOP/s (+ the reduce / bias / ReLU pass)" % (s, t, 2.0 * M * N * K / t * 1e-6))
