"""Time the fc6 dW tail variants at the bench shape: GEMM (bf16 out) + sgd_step vs drn_gemm_nt_sgd (fused)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
M, N, K = 2048, 50176, 2048
dev = "cuda"
A = (torch.randn((M, K), device=dev) * 0.05).to(torch.bfloat16)
B = (torch.randn((N, K), device=dev) * 0.5).to(torch.bfloat16)
w = torch.randn((M, N), device=dev)
mom = torch.zeros_like(w)
sh = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
g16 = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
seg[0] = (0, M * N, 0.01, 5e-4)
seg_dev = torch.from_numpy(seg.view(np.uint8)).to(dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def separate():
    ops.gemm_nt(A, B, M, N, K, out=g16.unsqueeze(0))
    ops.sgd_step(w.view(-1), mom.view(-1), g16.view(-1), seg_dev, 1, 0.9, False, shadow=sh.view(-1))


print("GEMM(bf16 out) then SGD : %7.1f us" % timeit(separate))
print("GEMM alone              : %7.1f us" % timeit(lambda: ops.gemm_nt(A, B, M, N, K, out=g16.unsqueeze(0))))
print("fused (stagger=%s)      : %7.1f us" % (os.environ.get("DRN_SGD_STAGGER", "0"),
                                              timeit(lambda: ops.gemm_nt_sgd(A, B, M, N, K, w, mom, sh, seg_dev, 0.9, False))))
