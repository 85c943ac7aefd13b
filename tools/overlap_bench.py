"""Does the fc6 dW GEMM overlap with the optimizer pass?  The tail of the step at the bench shape, stand-alone:

    main stream :  dW slab 0 ............ | dW slab 1 ............ | (next step's fc6 forward would start here)
    opt  stream :                           SGD slab 0 ...........   SGD slab 1 .........

timed from the first dW launch to the end of the last SGD launch, for the one-tile and the persistent 256x256 GEMM
and for several optimizer grids (workgroups that live for the whole launch; two 256-thread ones fit on a CU beside a
resident GEMM workgroup).  Also each piece alone, and everything on one stream (serial).
  python tools/overlap_bench.py [slabs]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
D1, K1, R = 2048, 50176, 2048
NSLAB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda"
torch.manual_seed(0)
dPT = (torch.randn((D1, R), device=dev) * 0.05).to(torch.bfloat16)
AT = (torch.randn((K1, R), device=dev) * 0.5).to(torch.bfloat16)
w = torch.randn((D1 * K1,), device=dev) * 0.02
mom = torch.randn_like(w) * 0.01
sh = torch.zeros((D1 * K1,), dtype=torch.bfloat16, device=dev)
g16 = torch.zeros((D1, K1), dtype=torch.bfloat16, device=dev)
rows = D1 // NSLAB
segs = []
for s in range(NSLAB):
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (s * rows * K1, rows * K1, 0.0, 5e-4)  # lr 0: the weights keep their scale over thousands of launches
    segs.append(torch.from_numpy(seg.view(np.uint8)).to(dev))
opt_stream = torch.cuda.Stream()


def dW(s):
    ops.gemm_nt(dPT[s * rows: (s + 1) * rows], AT, rows, K1, R, out=g16[s * rows: (s + 1) * rows].unsqueeze(0))


def sgd(s):
    ops.sgd_step(w, mom, g16.view(-1), segs[s], 1, 0.9, False, shadow=sh, grad_off=0)


def overlapped():
    main = torch.cuda.current_stream()
    for s in range(NSLAB):
        dW(s)
        ev = torch.cuda.Event()
        ev.record(main)
        opt_stream.wait_event(ev)
        with torch.cuda.stream(opt_stream):
            sgd(s)
    main.wait_stream(opt_stream)


def serial():
    for s in range(NSLAB):
        dW(s)
    for s in range(NSLAB):
        sgd(s)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("fc6 dW tail, %d slabs of %d rows; dW 411 GF, optimizer pass %.2f GB" % (NSLAB, rows, 20.0 * D1 * K1 / 1e9))
for persist in (0, 1):
    ops.tune(ops.TUNE_GEMM_PERSISTENT, persist)
    t = timeit(lambda: [dW(s) for s in range(NSLAB)])
    print("persistent=%d  dW alone (all slabs)            : %7.1f us  (%.0f TFLOP/s)" % (persist, t, 2.0 * D1 * K1 * 2000 / t / 1e6))
    for grid in (1024, 512, 256):
        ops.tune(ops.TUNE_SGD_GRID, grid)
        if persist == 0:
            t = timeit(lambda: [sgd(s) for s in range(NSLAB)])
            print("              SGD alone, grid %4d              : %7.1f us  (%.2f TB/s)" % (grid, t, 20.0 * D1 * K1 / t / 1e6))
        print("persistent=%d  SGD grid %4d  overlapped %7.1f us   serial %7.1f us" % (persist, grid, timeit(overlapped), timeit(serial)))
ops.tune(ops.TUNE_SGD_GRID, 512)
ops.tune(ops.TUNE_GEMM_PERSISTENT, 1)
