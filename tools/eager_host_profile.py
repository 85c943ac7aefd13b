"""cProfile of the eager training step's HOST side over the rotating real shapes."""
import cProfile
import io
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.argv = [sys.argv[0], "1"]
import runpy

ns = runpy.run_path(os.path.join(os.path.dirname(__file__), "eager_shapes_bench.py"))
step, batches, drain = ns["step"], ns["batches"], ns["drain"]
for label, idx in (("rotating shapes", lambda i: i),):  # (a fixed-shape phase would prefetch the batch that is in flight)
    drain()
    for i in range(4):
        step(idx(i))
    torch.cuda.synchronize()
    drain()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(16):
        step(idx(i))
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("==== %s: 16 steps ====" % label)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:48]))
