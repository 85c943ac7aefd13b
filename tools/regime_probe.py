"""What bounds the fc6 GEMMs inside the step - power / clocks, or the kernel?  Each launch is timed on its own (HIP
events) with an idle gap in front of it (torch.cuda._sleep): if the launch gets much faster with long gaps, the
back-to-back rate is set by the package power limit and the step's duty cycle matters; if not, by the kernel itself.
Also: the per-slab forward launch alone vs beside one optimizer pass of a row slab (the late-join overlap).
  python tools/regime_probe.py"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"
D1, K1, R = 2048, 50176, 2000
torch.manual_seed(0)
A = (torch.randn((R, K1), device=dev) * 0.5).to(torch.bfloat16)
W = (torch.randn((D1, K1), device=dev) * 0.02).to(torch.bfloat16)
part4 = torch.empty((4, R, D1), dtype=torch.float32, device=dev)
part8 = torch.empty((8, R, D1), dtype=torch.float32, device=dev)
dPT = (torch.randn((D1, 2048), device=dev) * 0.05).to(torch.bfloat16)
AT = (torch.randn((K1, 2048), device=dev) * 0.5).to(torch.bfloat16)
g16 = torch.zeros((D1, K1), dtype=torch.bfloat16, device=dev)
w = torch.randn((D1 * K1,), device=dev) * 0.02
mom = torch.randn_like(w) * 0.01
sh = torch.zeros((D1 * K1,), dtype=torch.bfloat16, device=dev)
seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
seg[0] = (1024 * K1, 1024 * K1, 0.0, 5e-4)
seg = torch.from_numpy(seg.view(np.uint8)).to(dev)


def fwd_full():
    ops.gemm_nt(A, W, R, D1, K1, out=part4, splits=4)


def fwd_half(h):
    ops.gemm_nt(A, W[h * 1024:(h + 1) * 1024], R, 1024, K1, out=part8[:, :, h * 1024:(h + 1) * 1024], splits=8)


def dw_slab():
    ops.gemm_nt(dPT[:1024], AT, 1024, K1, 2048, out=g16[:1024].unsqueeze(0))


def sgd():
    ops.sgd_step(w, mom, g16.view(-1), seg, 1, 0.9, False, shadow=sh, grad_off=0)


def gap_timed(fn, gap_cycles, n=60):
    """mean duration of fn's launch when every launch is preceded by an idle gap"""
    evs = []
    for _ in range(n):
        if gap_cycles:
            torch.cuda._sleep(gap_cycles)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in evs[n // 2:]]
    return sum(t) / len(t)


for name, fn, gf in (("fc6 forward, one launch (4 splits)", fwd_full, 411.04), ("fc6 forward, columns 0:1024 (8 splits)", lambda: fwd_half(0), 205.52),
                     ("fc6 dW row slab 0:1024", dw_slab, 205.52), ("sgd row slab (1.03 GB)", sgd, 0)):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    row = []
    for gap_us in (0, 100, 400, 1600, 6400):
        t = gap_timed(fn, int(gap_us * 100))  # _sleep counts ~100 MHz ticks on MI300-class parts; the gap is printed as asked, not measured
        row.append("%5d:%7.1f" % (gap_us, t) + (" (%4.0f TF)" % (gf / t * 1e3) if gf else ""))
    print("%-42s gap_us:us  %s" % (name, "  ".join(row)))

# the late-join overlap: forward columns 0:1024 beside the optimizer pass of rows 1024:2048
opt = torch.cuda.Stream()


def both():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main); opt.wait_event(ev)
    with torch.cuda.stream(opt):
        sgd()
    fwd_half(0)
    main.wait_stream(opt)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("forward half alone %.1f us, sgd slab alone %.1f us, both on two streams %.1f us, one after the other %.1f us"
      % (timeit(lambda: fwd_half(0)), timeit(sgd), timeit(both), timeit(lambda: (fwd_half(0), sgd()))))
print("two forward halves %.1f us vs one full launch %.1f us" % (timeit(lambda: (fwd_half(0), fwd_half(1))), timeit(fwd_full)))
