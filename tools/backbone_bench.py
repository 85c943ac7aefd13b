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
HW = os.environ.get("BB_HW")
if HW:  # e.g. BB_HW=800,1216: a realistic training / test-time scale instead of the 224x224 benchmark image
    H, W = [int(x) for x in HW.split(",")]
    for b in batches:
        b[0]["image"] = torch.randint(0, 256, (3, H, W)).float().cuda()
    # R50-C4 trunk: 7.90 GFLOP at 224x224, linear in the pixel count
    print("trunk FLOPs at %dx%d: %.1f GF" % (H, W, 7.90 * H * W / (224 * 224)))


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
t = a.elapsed_time(b) / 20 * 1e3
print("graph  : %.1f us per trunk forward" % t)
if HW:
    print("        = %.1f TFLOP/s" % (7.90 * H * W / (224 * 224) / t * 1e3))
