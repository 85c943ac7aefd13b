"""The eager training step on ONE real-size shape, for rocprofv3 (kernel timeline per stream) and for timing with the trunk
prefetch on / off.   EAGER_HWR=1000,1464,1947 PREFETCH=1 python tools/eager_one_shape.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench as B
from __graft_entry__ import load_package

pkg = load_package()
pkg._cabi.lib()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.engine import DataParallel, build_optimizer
from drn_wsod_pytorch_amd.modeling import build_model
from drn_wsod_pytorch_amd.structures import Boxes, Instances

dev = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, W, R = [int(x) for x in os.environ.get("EAGER_HWR", "1000,1464,1947").split(",")]
prefetch = os.environ.get("PREFETCH", "1") == "1"
torch.manual_seed(1)
cfg = B.build_cfg(pkg, dev)
model = build_model(cfg)
B.init_weights(model, seed=0)
model.train()
opt = build_optimizer(cfg, model)
dp = DataParallel(model)
opt.enable_pipelined(dp)
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES


def batch(seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (3, H, W), generator=g).float()
    x0, y0 = torch.rand(R, generator=g) * (W - 40), torch.rand(R, generator=g) * (H - 40)
    bw, bh = 20 + torch.rand(R, generator=g) * (W - x0 - 20), 20 + torch.rand(R, generator=g) * (H - y0 - 20)
    boxes = torch.stack([x0, y0, (x0 + bw).clamp(max=W), (y0 + bh).clamp(max=H)], 1)
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(boxes.to(dev))
    prop.objectness_logits = torch.sort(torch.rand(R, generator=g), descending=True).values.to(dev)
    inst = Instances((H, W))
    inst.gt_boxes = Boxes(boxes[:2].clone())
    inst.gt_classes = torch.randperm(K, generator=g)[:2].to(torch.int64)
    return [{"image": img.to(dev), "proposals": prop, "instances": inst, "height": H, "width": W}]


batches = [batch(10 + i) for i in range(4)]


EARLY = os.environ.get("PREFETCH_EARLY", "0") == "1"  # A/B: prefetch in front of the forward (round 1's order)


def step(i):
    if prefetch and EARLY:
        model.prefetch_features(batches[(i + 1) % 4])
    losses = model(batches[i % 4])
    if prefetch and not EARLY:  # Trainer.run_step's order: the next batch's trunk + pooling between forward and backward
        model.prefetch_features(batches[(i + 1) % 4])
    if not model.backward_losses(1.0):
        sum(losses.values()).backward()
    dp.finish()
    opt.step(dp.grad_scale)
    opt.zero_grad()


for i in range(6):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6, 6 + steps):
    step(i)
te = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d x %d, R = %d, trunk prefetch %s: wall %.3f ms/step, host enqueue %.3f ms/step" %
      (H, W, R, "on" if prefetch else "off", dt / steps * 1e3, te / steps * 1e3))
