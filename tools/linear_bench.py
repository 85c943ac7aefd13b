"""fc7 forward of the bench workload - H2 = dropout(relu(H1 [2000 x 2048] . W2 [4096 x 2048]^T + b)) with its transposed copy - as
the step runs it today (drn_gemm_nt with split-K partials + drn_bias_act_fwd) against ONE launch of the eight-wave kernel
(drn_linear_act_fwd, pp8.hip) at ring depths 3 / 4 / 5; each form replayed 20x from a hipGraph.
  python tools/linear_bench.py [M N K]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
M, N, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2000, 4096, 2048)
dev = "cuda"
A = (torch.randn((M, K), device=dev) * 0.5).to(dt)
W = (torch.randn((N, K), device=dev) * 0.03).to(dt)
bias = torch.randn(N, device=dev) * 0.1
Mp = ops.kpad(M, dt)
out = torch.zeros((M, N), dtype=dt, device=dev)
outT = torch.zeros((N, Mp), dtype=dt, device=dev)
seed_dev = torch.zeros((1,), dtype=torch.int64, device=dev)


def timed(f):
    for _ in range(3):
        f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3


gf = 2.0 * M * N * K / 1e9
print("linear + bias + ReLU + dropout + transposed copy, [%d x %d] . [%d x %d]^T = %.1f GF" % (M, K, N, K, gf))
for s in (1, 2, 4):
    part = torch.empty((s, M, N), dtype=torch.float32, device=dev)

    def two():
        ops.gemm_nt(A, W, M, N, K, out=part, splits=s)
        ops.bias_act_fwd(part, M, N, bias, True, None, 77, 0.5, out=out, outT=outT, seed_dev=seed_dev)

    t = timed(two)
    tg = timed(lambda: ops.gemm_nt(A, W, M, N, K, out=part, splits=s))
    print("  gemm_nt splits=%d + bias_act_fwd: %6.1f us (GEMM alone %5.1f us = %4.0f TFLOP/s)" % (s, t, tg, gf / tg * 1e3))
ref = out.clone()
for st in (3, 4, 5):
    ops.tune(ops.TUNE_PP8_STAGES, st)
    for want_t in (True, False):
        t = timed(lambda: ops.linear_act_fwd(A, W, M, N, K, bias, True, None, 77, 0.5, out=out, outT=outT if want_t else None,
                                             seed_dev=seed_dev))
        print("  linear_act_fwd, %d stages%s: %6.1f us = %4.0f TFLOP/s" % (st, "" if want_t else " (no transposed copy)", t, gf / t * 1e3))
ops.tune(ops.TUNE_PP8_STAGES, 4)
part = torch.empty((1, M, N), dtype=torch.float32, device=dev)
ops.gemm_nt(A, W, M, N, K, out=part, splits=1)
ops.bias_act_fwd(part, M, N, bias, True, None, 77, 0.5, out=ref, seed_dev=seed_dev)
print("  bit-identical to splits=1 + bias_act_fwd:", bool(torch.equal(ref, out)))
