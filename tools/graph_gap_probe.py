"""What a transition between pieces of the step costs on the GPU (the step is graph | eager GEMM | graph | eager GEMM | graph
today): the same four ~50-us GEMM launches issued 200 times as (a) one hipGraph, (b) graph + eager + graph + eager, (c) four
graphs, (d) all eager; and whether a timing event can be recorded INSIDE a captured graph (torch.cuda.Event(external=True))."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dt = torch.bfloat16
dev = "cuda"
M, N, K = 2048, 4096, 2048
A = (torch.randn((M, K), device=dev) * 0.5).to(dt)
W = (torch.randn((N, K), device=dev) * 0.03).to(dt)
outs = [torch.empty((1, M, N), dtype=torch.float32, device=dev) for _ in range(4)]
fs = [(lambda o=o: ops.gemm_nt(A, W, M, N, K, out=o)) for o in outs]
for f in fs:
    f()
torch.cuda.synchronize()


def graph_of(funcs):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in funcs:
            f()
    return g


def timed(step, reps=200):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


g_all = graph_of(fs)
g1 = [graph_of([f]) for f in fs]
print("four GEMM launches per step [%d x %d x %d]" % (M, N, K))
print("  (a) one graph of four:            %7.1f us / step" % timed(lambda: g_all.replay()))
print("  (b) graph, eager, graph, eager:   %7.1f us / step" % timed(lambda: (g1[0].replay(), fs[1](), g1[2].replay(), fs[3]())))
print("  (c) four graphs of one:           %7.1f us / step" % timed(lambda: [g.replay() for g in g1]))
print("  (d) four eager launches:          %7.1f us / step" % timed(lambda: [f() for f in fs]))
g2 = [graph_of(fs[:2]), graph_of(fs[2:])]
print("  (e) two graphs of two:            %7.1f us / step" % timed(lambda: [g.replay() for g in g2]))
try:
    e0 = torch.cuda.Event(enable_timing=True, external=True)
    e1 = torch.cuda.Event(enable_timing=True, external=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fs[0]()
        e0.record()
        fs[1]()
        e1.record()
        fs[2]()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("  external timing events inside a captured graph: GEMM 2 took %.1f us" % (e0.elapsed_time(e1) * 1e3))
except Exception as ex:  # noqa: BLE001
    print("  external timing events inside a captured graph: NOT available (%s: %s)" % (type(ex).__name__, str(ex)[:200]))
