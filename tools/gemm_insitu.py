"""Why is the fc6 forward GEMM slower inside the step than back-to-back?  Time it (HIP events around the launch only)
(a) back-to-back, (b) after a 1.6 GB streaming pass that evicts L2 / Infinity Cache (what the SGD pass does in the
step), (c) after ~1 ms of idle-ish tiny kernels (clock ramp), (d) with A freshly rewritten before each launch."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
M, N, K, S = 2000, 2048, 50176, 4
A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
B = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty((S, M, N), dtype=torch.float32, device="cuda")
big = torch.empty((400 * 1024 * 1024,), dtype=torch.float32, device="cuda")
tiny = torch.zeros((64,), device="cuda")


def run(pre, n=12):
    ts = []
    for i in range(n + 2):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.gemm_nt(A, B, M, N, K, out=out, splits=S)
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(a.elapsed_time(b) * 1e3)
    return sum(ts) / len(ts), min(ts), max(ts)


def evict():
    big.add_(1.0)


def idle():
    for _ in range(150):
        tiny.add_(1.0)


def rewrite():
    A.mul_(1.0)


for name, pre in [("back-to-back", lambda: None), ("after 3.2 GB streaming pass", evict), ("after 150 tiny kernels", idle),
                  ("A rewritten just before", rewrite)]:
    print("%-30s avg %6.1f us  min %6.1f  max %6.1f" % ((name,) + run(pre)))
