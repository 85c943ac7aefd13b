"""Can a conv workgroup share a CU with a 256x256 GEMM workgroup?  A chain of 10 dependent res4-sized 1x1 convs
(64x64 tiles, 64 workgroups each) is launched on a side stream while the fc6 forward GEMM (256 workgroups, one per CU,
~340 us) runs on the main stream: if the conv workgroups are co-resident the chain finishes during the GEMM, otherwise
only after it."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
from drn_wsod_pytorch_amd import ops

dev = "cuda"
bf = torch.bfloat16
M, N, K = 2000, 2048, 50176
A = (torch.randn(M, K, device=dev) * 0.01).to(bf)
B = (torch.randn(N, K, device=dev) * 0.01).to(bf)
x = (torch.randn(1, 14, 14, 1024, device=dev)).to(bf)
w = (torch.randn(1024, 1024, device=dev) * 0.03).to(bf)  # 1x1 conv 1024 -> 1024, packed [Cout, Cin]
side = torch.cuda.Stream()


def gemm():
    return ops.gemm_nt(A, B, M, N, K, splits=4)


def chain(n=10):
    y = x
    for _ in range(n):
        y = ops.conv2d_nhwc(y, w, 1024, 1, 1, relu=True)
    return y


for _ in range(3):
    gemm()
    chain()
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
# chain alone
a, b = ev(), ev()
a.record()
chain()
b.record()
torch.cuda.synchronize()
print("chain of 10 convs alone      : %.1f us" % (a.elapsed_time(b) * 1e3))
a, b = ev(), ev()
a.record()
gemm()
b.record()
torch.cuda.synchronize()
print("fc6 forward GEMM alone       : %.1f us" % (a.elapsed_time(b) * 1e3))
for rep in range(3):
    g0, g1, c0, c1 = ev(), ev(), ev(), ev()
    torch.cuda.synchronize()
    g0.record()
    gemm()
    g1.record()
    with torch.cuda.stream(side):
        c0.record(side)
        chain()
        c1.record(side)
    torch.cuda.synchronize()
    print("together: GEMM %.1f us | chain %.1f us (start +%.1f us after the GEMM's start, end %+.1f us vs the GEMM's end)"
          % (g0.elapsed_time(g1) * 1e3, c0.elapsed_time(c1) * 1e3, g0.elapsed_time(c0) * 1e3, g1.elapsed_time(c1) * 1e3))
