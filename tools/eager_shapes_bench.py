"""The eager training step on VOC-shaped inputs whose size changes EVERY step (what real data looks like: multi-scale
shortest edge 480-1200, aspect ratios 0.5-2, 500-2000 proposals after filtering) - the case GraphedTrainStep refuses.
Reports wall time per step, host enqueue time per step and GPU-busy time per step (HIP events), i.e. whether this
path is host-bound at real image sizes.  Same model / optimizer / pipelined SGD as bench.py; the next batch's frozen
trunk is prefetched on the side stream (model.prefetch_features) like in bench.py's eager step.
  python tools/eager_shapes_bench.py [steps]      (WORKLOAD=r50dc5: the shipped DC5 recipe instead of the constructed R50-C4 model)"""
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = B.build_cfg(pkg, dev)
if os.environ.get("WORKLOAD", "r50c4") == "r50dc5":  # the shipped recipe's trunk (oicr_WSR_50_DC5_1x.yaml): res4 / res5 dilated at stride 8
    cfg.merge_from_list(["MODEL.RESNETS.OUT_FEATURES", "['res5']", "MODEL.ROI_HEADS.IN_FEATURES", "['res5']", "MODEL.RESNETS.RES5_DILATION", "2"])
model = build_model(cfg)
B.init_weights(model, seed=0)
model.train()
opt = build_optimizer(cfg, model)
dp = DataParallel(model)
opt.enable_pipelined(dp, fused_tn={"0": False, "1": True}.get(os.environ.get("FUSED_TN", ""), None))  # FUSED_TN=0/1: A/B of the fused fc6 dW + SGD launch
K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
torch.manual_seed(int(os.environ.get("SEED", "1")))  # (the dropout masks are keyed by torch's seed: fixed, so that runs compare)
for kv in filter(None, os.environ.get("TUNE", "").split(",")):  # TUNE=19=2,...: drn_tune knobs for A/B runs
    from drn_wsod_pytorch_amd import ops as _ops
    _ops.tune(int(kv.split("=")[0]), int(kv.split("=")[1]))


def batch(seed, H, W, R):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (3, H, W), generator=g).float()
    x0, y0 = torch.rand(R, generator=g) * (W - 40), torch.rand(R, generator=g) * (H - 40)
    bw, bh = 20 + torch.rand(R, generator=g) * (W - x0 - 20), 20 + torch.rand(R, generator=g) * (H - y0 - 20)
    boxes = torch.stack([x0, y0, (x0 + bw).clamp(max=W), (y0 + bh).clamp(max=H)], 1)
    prop = Instances((H, W))
    prop.proposal_boxes = Boxes(boxes.to(dev))
    prop.objectness_logits = torch.sort(torch.rand(R, generator=g), descending=True).values.to(dev)
    inst = Instances((H, W))
    G = int(torch.randint(1, 4, (1,), generator=g))
    inst.gt_boxes = Boxes(boxes[:G].clone())
    inst.gt_classes = torch.randperm(K, generator=g)[:G].to(torch.int64)
    return [{"image": img.to(dev), "proposals": prop, "instances": inst, "height": H, "width": W}]


# WSL configs: INPUT.MIN_SIZE_TRAIN (480, 576, 688, 864, 1000, 1200), MAX_SIZE_TRAIN 2000; VOC aspect ratios
g = torch.Generator().manual_seed(7)
shapes = []
for i in range(16):
    short = (480, 576, 688, 864, 1000, 1200)[int(torch.randint(0, 6, (1,), generator=g))]
    ar = float(0.6 + torch.rand(1, generator=g) * 1.2)
    H, W = (short, int(short * ar)) if ar >= 1 else (int(short / ar), short)
    R = int(torch.randint(500, 2001, (1,), generator=g))
    shapes.append((H, W, R))
batches = [batch(100 + i, *s) for i, s in enumerate(shapes)]
#  the weight scales of bench.init_weights belong to the 224 x 224 / C4 workload: calibrate fc6 on a real-size batch of
# THIS trunk, and check the losses before a time is printed
print("fc6 calibration factor on %d x %d: %.3f" % (shapes[0][0], shapes[0][1], B.calibrate_fc6(model, batches[0])))
opt.zero_grad()


def step(i):
    losses = model(batches[i % len(batches)])
    model.prefetch_features(batches[(i + 1) % len(batches)])
    if os.environ.get("AUTOGRAD", "0") == "1" or not model.backward_losses(1.0):  # Trainer.run_step's order
        sum(losses.values()).backward()
    dp.finish()
    opt.step(dp.grad_scale)
    opt.zero_grad()
    return losses


def drain():
    """drop a prefetched batch that will not be consumed (its fc6-operand set goes back to the engine)"""
    for v in getattr(model, "_prefetch_cache", {}).values():
        if v[3] is not None:
            v[3]["pooled"]["state"] = "free"
    model._prefetch_cache = {}
    torch.cuda.synchronize()


nw = len(batches) + 2
for i in range(nw):  # every shape once: workspaces, packed weights, allocator
    last = step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(nw, nw + steps):
    last = step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pix = sum(s[0] * s[1] for s in shapes) / len(shapes)
print("eager step, %d distinct (H, W, R) in rotation (mean %.0f x %.0f pixels, mean R %.0f):" %
      (len(shapes), pix ** 0.5, pix ** 0.5, sum(s[2] for s in shapes) / len(shapes)))
sane = B.assert_sane_losses(last, "eager rotation (%s)" % os.environ.get("WORKLOAD", "r50c4"))
print("  wall %.3f ms/step (%.1f img/s)   host enqueue %.3f ms/step   losses %s" %
      (dt / steps * 1e3, steps / dt, t_enq / steps * 1e3, {k: round(v, 4) for k, v in sane.items()}))
if os.environ.get("JSON", "0") == "1":  # bench.py's side_real_shapes child
    import json
    print(json.dumps({"workload": os.environ.get("WORKLOAD", "r50c4"), "ms_per_step": dt / steps * 1e3, "host_enqueue_ms_per_step": t_enq / steps * 1e3,
                      "steps": steps, "shapes": len(shapes), "mean_pixels": pix, "losses": sane}))
    sys.exit(0)
# the same shapes one by one: eager vs host time
for (H, W, R), b in list(zip(shapes, batches))[:6]:
    drain()
    def one(i):
        losses = model(b)
        model.prefetch_features(b)
        if os.environ.get("AUTOGRAD", "0") == "1" or not model.backward_losses(1.0):
            sum(losses.values()).backward()
        dp.finish(); opt.step(dp.grad_scale); opt.zero_grad()
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        one(i)
    te = time.perf_counter() - t0
    torch.cuda.synchronize()
    d = time.perf_counter() - t0
    print("  %4d x %4d, R = %4d : wall %.3f ms/step, host enqueue %.3f ms/step" % (H, W, R, d / 20 * 1e3, te / 20 * 1e3))
