"""Time the frozen R50-C4 trunk (preprocess + 45 conv launches + pools) on its own: eager and as a replayed hipGraph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from __graft_entry__ import load_package

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.modeling import build_model

cfg = bench.build_cfg(pkg, "cuda")
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
batches = bench.synthetic_batches(2, 2000, 20, "cuda", 0, pkg)


def fwd():
    with torch.no_grad():
        imgs = model.preprocess_image(batches[0])
        return model.backbone(imgs.tensor)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fwd()
b.record()
torch.cuda.synchronize()
print("eager  : %.1f us per trunk forward" % (a.elapsed_time(b) / 20 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd()
g.replay()
torch.cuda.synchronize()
a.record()
for _ in range(20):
    g.replay()
b.record()
torch.cuda.synchronize()
print("graph  : %.1f us per trunk forward" % (a.elapsed_time(b) / 20 * 1e3))
