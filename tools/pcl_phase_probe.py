"""Phase clocks of the PCL kernel inside the bench workload (library built with -DPCL_PROFILE): runs eager training
steps of bench.py's PCL model and prints, per step, the last branch's phase ticks, centres and greedy iterations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.engine import build_optimizer  # noqa: E402
from drn_wsod_pytorch_amd.modeling import build_model  # noqa: E402

cfg = bench.build_cfg(pkg, "cuda")
cfg.merge_from_list(["MODEL.ROI_HEADS.NAME", "PCLROIHeads"])
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.train()
opt = build_optimizer(cfg, model)
batches = bench.synthetic_batches(8, 2000, 20, "cuda", 0, pkg)
names = ["gather+sort", "prefix+cuts", "search", "members", "greedy", "top5", "assign", "loss"]
for i in range(int(os.environ.get("STEPS", 12))):
    opt.zero_grad()
    losses = model(batches[i % 8])
    sum(losses.values()).backward()
    opt.step()
    tg = model.roi_heads._last_state["aux"]["targets"]
    t = tg[-1]["pc_scores"][-10:].cpu().numpy()
    print(i, "G=%d" % int(batches[i % 8][0]["instances"].gt_classes.numel()), "centres", [int(x["n_pc"]) for x in tg],
          {n: int(v) for n, v in zip(names, t[:8])}, "iters", int(t[8]), "fg", [int((x["labels"] > 0).sum()) for x in tg])
