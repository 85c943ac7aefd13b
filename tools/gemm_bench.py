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


def sustained(name, M, N, K, dtype, splits, seconds=1.2):
    """the same launch repeated for ~1 s: the package reaches its 1400 W cap and the shader clock drops to ~1.77 GHz,
    which is the regime the GEMMs see inside the training step (rocm-smi); report the rate of the second half"""
    kp = ops.kpad(K, dtype)
    A = (torch.randn((M, kp), device="cuda") * 0.5).to(dtype)
    B = (torch.randn((N, kp), device="cuda") * 0.05).to(dtype)
    out = torch.empty((splits, M, N), dtype=torch.float32, device="cuda")
    n = 200
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(n):
        ops.gemm_nt(A, B, M, N, kp, out=out, splits=splits)
    torch.cuda.synchronize()
    t1 = 0.0
    reps = 0
    import time
    t0 = time.time()
    while time.time() - t0 < seconds:
        e[0].record()
        for _ in range(n):
            ops.gemm_nt(A, B, M, N, kp, out=out, splits=splits)
        e[1].record()
        torch.cuda.synchronize()
        t1 = e[0].elapsed_time(e[1]) / n
        reps += 1
    print("%-28s sustained (%d x %d launches): %.3f ms  %.1f TFLOP/s" % (name, reps, n, t1, 2.0 * M * N * K / t1 / 1e9))


if __name__ == "__main__":
    bf, f32 = torch.bfloat16, torch.float32
    if len(sys.argv) > 1 and sys.argv[1] == "sustained":
        for tile in [int(x) for x in sys.argv[2:]] or [256]:
            ops.gemm_set_tile(tile)
            print("---- tile", tile)
            sustained("fc6 fwd", 2000, 2048, 50176, bf, 4)
            sustained("fc6 dW", 2048, 50176, 2048, bf, 1)
            sustained("square 8192", 8192, 8192, 8192, bf, 1)
        sys.exit(0)
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
