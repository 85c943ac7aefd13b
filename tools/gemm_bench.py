"""Micro-benchmark of drn_gemm_nt on the BASELINE configs[1] shapes (HIP-event timed)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")


def bench(name, M, N, K, dtype, splits, iters=10):
    dev = "cuda"
    A = (torch.randn((M, K), device=dev) * 0.5).to(dtype)
    B = (torch.randn((N, K), device=dev) * 0.05).to(dtype)
    out = torch.empty((splits, M, N), dtype=torch.float32, device=dev)
    for _ in range(2):
        ops.gemm_nt(A, B, M, N, K, out=out, splits=splits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt(A, B, M, N, K, out=out, splits=splits)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print("%-28s %-8s M=%d N=%d K=%d splits=%d  %.3f ms  %.1f TFLOP/s" % (name, str(dtype).split('.')[-1], M, N, K, splits, ms, tf))


if __name__ == "__main__":
    bf, f32 = torch.bfloat16, torch.float32
    for tile in (255, 256):
        ops.gemm_set_tile(tile)
        print("---- tile", tile)
        for s in (2, 4, 8):
            bench("fc6 fwd", 2000, 2048, 50176, bf, s)
        bench("fc6 dW", 2048, 50176, 2048, bf, 1)
        bench("fc7 fwd", 2000, 4096, 2048, bf, 2)
        bench("fc7 dW", 4096, 2048, 2048, bf, 1)
        bench("square 4096", 4096, 4096, 4096, bf, 1)
        bench("square 8192", 8192, 8192, 8192, bf, 1)
        bench("fc6 fwd f32", 2000, 2048, 50176, f32, 4, iters=3)
    ops.gemm_set_tile(0)
    print("---- heuristic")
    for s in (1, 2, 4, 8):
        bench("fc6 fwd", 2000, 2048, 50176, bf, s)
    bench("fc6 dW", 2048, 50176, 2048, bf, 1)
    bench("fc7 fwd", 2000, 4096, 2048, bf, 1)
    bench("fc7 dW", 4096, 2048, 2048, bf, 1)
    bench("fc7 dX", 2000, 2048, 4096, bf, 1)
    bench("heads fwd", 2000, 103, 4096, bf, 8)
    bench("square 4096", 4096, 4096, 4096, bf, 1)
    bench("square 8192", 8192, 8192, 8192, bf, 1)
    bench("fc6 fwd f32", 2000, 2048, 50176, f32, 4, iters=3)
    bench("square 4096 f32", 4096, 4096, 4096, f32, 1, iters=3)
