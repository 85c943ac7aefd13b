"""Pick the thread count for bench.py's cpu_baseline: time one oracle train step (R50-C4, R=2000) per setting."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from __graft_entry__ import load_package

pkg = load_package()
from oracle import wsod_oracle as O

batches = bench.synthetic_batches(3, 2000, 20, "cpu", 0, pkg)
cfg = O.OracleCfg(arch="wsr50", out_feature="res4", res5_dilation=1, dropout=0.5)
p = O.init_params(cfg, seed=0)
opt = O.SGDState(cfg)
print("cpu_count", os.cpu_count())
for th in [int(x) for x in sys.argv[1:]]:
    torch.set_num_threads(th)
    ts = []
    for i in range(2):
        t0 = time.perf_counter()
        O.train_step(p, [batches[i][0]["_cpu"]], cfg, opt)
        ts.append(time.perf_counter() - t0)
    print("threads %3d: first %.1f s, second %.1f s" % (th, ts[0], ts[1]), flush=True)
