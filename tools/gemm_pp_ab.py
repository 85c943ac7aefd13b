"""Interleaved A/B of the two mainloops of the bf16 256x256 GEMM kernels (DRN_TUNE_GEMM_PINGPONG 0 / 1) on the fc6
shapes and a square, HIP-event timed, N rounds in ONE process (cdna_hip_programming.md rule 24), random operands."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
bf = torch.bfloat16
SHAPES = [("fc6 fwd", 2000, 2048, 50176, 4, False), ("fc6 dW slab", 1024, 49152, 2048, 1, True),
          ("fc7 fwd", 2000, 4096, 2048, 1, False), ("square 4096", 4096, 4096, 4096, 1, False),
          ("square 8192", 8192, 8192, 8192, 1, False)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
VARS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
n = 40
ops.gemm_set_tile(256)
for name, M, N, K, S, b16 in SHAPES:
    A = (torch.randn((M, K), device="cuda") * 0.5).to(bf)
    B = (torch.randn((N, K), device="cuda") * 0.05).to(bf)
    out = torch.empty((S, M, N), dtype=bf if b16 else torch.float32, device="cuda")
    res = {v: [] for v in VARS}
    for r in range(rounds):
        for pp in VARS:
            ops.tune(ops.TUNE_GEMM_PINGPONG, pp)
            for _ in range(5):
                ops.gemm_nt(A, B, M, N, K, out=out, splits=S)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.gemm_nt(A, B, M, N, K, out=out, splits=S)
            e1.record()
            torch.cuda.synchronize()
            res[pp].append(e0.elapsed_time(e1) / n)
    for pp in VARS:
        t = sorted(res[pp])
        med = t[len(t) // 2]
        print("%-12s pp=%d  median %.1f us (min %.1f)  %.0f TFLOP/s" % (name, pp, med * 1e3, t[0] * 1e3, 2.0 * M * N * K / med / 1e9))
ops.tune(ops.TUNE_GEMM_PINGPONG, 1)
# round 3: the fc6 dW slab with its second operand K-major (drn_gemm_tn reads the pooled matrix A [R][C*49] itself)
# against the NT form on a materialised A^T, interleaved
M, N, K, R = 1024, 49152, 2048, 2000
A = (torch.randn((M, K), device="cuda") * 0.5).to(bf)
A[:, R:] = 0
Bt = (torch.randn((R, 50176), device="cuda") * 0.05).to(bf)
B = torch.zeros((50176, K), dtype=bf, device="cuda")
B[:, :R] = Bt.t()
o1 = torch.empty((1, M, N), dtype=bf, device="cuda")
o2 = torch.empty((1, M, N), dtype=bf, device="cuda")
fns = {"NT": lambda: ops.gemm_nt(A, B[:N], M, N, K, out=o1), "TN": lambda: ops.gemm_tn(A, Bt[:, :N], M, N, K, R, out=o2)}
res = {k: [] for k in fns}
for r in range(rounds):
    for k, f in fns.items():
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / n)
print("fc6 dW slab, outputs equal:", bool(torch.equal(o1, o2)))
for k in fns:
    t = sorted(res[k])
    med = t[len(t) // 2]
    print("fc6 dW slab  %s  median %.1f us (min %.1f)  %.0f TFLOP/s" % (k, med * 1e3, t[0] * 1e3, 2.0 * M * N * K / med / 1e9))
