"""Trainable trunk (MODEL.BACKBONE.FREEZE_AT = 2) at the bench shape: eager step vs GraphedFullStep (ms per step)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench as B
from __graft_entry__ import load_package

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.engine import GraphedFullStep, build_optimizer
from drn_wsod_pytorch_amd.modeling import build_model

freeze = int(os.environ.get("FREEZE_AT", "2"))
cfg = B.build_cfg(pkg, "cuda:0")
cfg.merge_from_list(["MODEL.BACKBONE.FREEZE_AT", str(freeze)])
for mode in ("eager", "graph"):
    model = build_model(cfg)
    B.init_weights(model, seed=0)
    model.train()
    opt = build_optimizer(cfg, model)
    batches = B.synthetic_batches(4, 2000, 20, "cuda:0", 0, pkg)
    stepper = GraphedFullStep(model, opt, batches[0]) if mode == "graph" else None

    def step(i):
        b = batches[i % 4]
        if stepper is not None:
            return stepper.step(b)
        opt.zero_grad()
        losses = model(b)
        sum(losses.values()).backward()
        opt.step()
        return losses

    for i in range(5):
        last = step(i)
        print("   %-5s step %d  %s" % (mode, i, {k: round(float(v), 5) for k, v in last.items()}))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        last = step(5 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print("FREEZE_AT=%d %-5s  %.3f ms/step  %.1f img/s  losses %s" % (freeze, mode, dt, 1e3 / dt,
                                                                       {k: round(float(v), 4) for k, v in last.items()}))
    del model, opt, stepper
    torch.cuda.empty_cache()
