"""CSCROIHeads at the shipped shape (csc_WSR_18_DC5_1x.yaml: WS-ResNet18 DC5, 4096-wide neck, 20 classes, one image per
GPU; 688 x 917 image, 2000 proposals), eager steps (the head reads K scores on the host each step):
  * step without class maps (past CSC_MAX_ITER / no class reaches tau: what a random-init model does at tau = 0.7),
  * step with N class maps (tau = 0: every labelled class gets its image-gradient pass),
  * one image-gradient pass alone (seed -> heads d/dx -> RoIPool backward -> trunk d/dx -> map -> table -> CSCPool).
  python tools/csc_bench.py [--precision bf16|fp32] [--arch wsr18|vgg16] [--classes 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import golden_util as G  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from oracle import wsod_oracle as O  # noqa: E402  (synthetic inputs only)

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--arch", default="wsr18")
ap.add_argument("--classes", type=int, default=2)
ap.add_argument("--H", type=int, default=688)
ap.add_argument("--W", type=int, default=917)
ap.add_argument("--R", type=int, default=2000)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
load_package()
from drn_wsod_pytorch_amd.engine import build_optimizer  # noqa: E402

if args.arch == "vgg16":
    ocfg = O.OracleCfg(arch="vgg16", out_feature="plain5", res5_dilation=2, dan_dim=(4096, 4096), num_classes=20,
                       pixel_mean=(103.939, 116.779, 123.68), heads="csc", refine_num=0, refine_reg=(), mean_loss=False,
                       base_lr=1e-5)
else:
    ocfg = O.OracleCfg(arch="wsr18", out_feature="res5", res5_dilation=1, res2_out=64, dan_dim=(4096, 4096),
                       num_classes=20, heads="csc", refine_num=0, refine_reg=(), mean_loss=False, base_lr=1e-5)
batch = O.synthetic_batch(1, args.R, ocfg, seed=3, H=args.H, W=args.W)
batch[0]["gt_classes"] = torch.arange(args.classes)
ocfg.base_lr = 0.0  # the weights stay where the seeded init put them: every timed step does the same work
cfg, model = G.drn_model(ocfg, 3, "cuda", 5, args.precision)
model.train()
opt = build_optimizer(cfg, model)
inputs = G.drn_inputs(batch)


def run(n, tau, it):
    model.roi_heads.tau = tau
    for i in range(n):
        model.roi_heads.iter = it
        opt.zero_grad()
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
    torch.cuda.synchronize()


def timed(tau, it):
    run(3, tau, it)
    t0 = time.perf_counter()
    run(args.steps, tau, it)
    return (time.perf_counter() - t0) / args.steps * 1e3


t_plain = timed(0.7, 10 ** 9)
t_maps = timed(0.0, 1)
aux = model.roi_heads._last_state["aux"]
nz = int((aux["cpgs"].flatten(1).amax(1) > 0).sum())
print("CSC %s %s  image %dx%d  R=%d  K=20: step without maps %.2f ms, with %d class maps %.2f ms -> %.2f ms per "
      "image-gradient pass (maps non-zero: %d; W range %.3f .. %.3f; losses %s)" % (
          args.arch, args.precision, args.H, args.W, args.R, t_plain, args.classes, t_maps,
          (t_maps - t_plain) / max(1, args.classes), nz, float(aux["W"].min()), float(aux["W"].max()),
          {k: round(float(v), 4) for k, v in model(inputs).items()}))
