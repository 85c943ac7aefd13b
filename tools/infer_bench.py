"""Inference (eval) pass of R50-C4 at TTA-like image sizes: wall time per image and the top kernels.
INFER_SIZES="688x920,224x224" picks the sizes, WORKLOAD=r50dc5 the shipped DC5 recipe, INFER_PLAN=0 runs the trunk layer by layer instead of through its launch plan."""
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
from drn_wsod_pytorch_amd.structures import Boxes, Instances

cfg = bench.build_cfg(pkg, "cuda")
if os.environ.get("WORKLOAD", "r50c4") == "r50dc5":  # the shipped recipe's trunk (oicr_WSR_50_DC5_1x.yaml): res4 / res5 dilated at stride 8
    cfg.merge_from_list(["MODEL.RESNETS.OUT_FEATURES", "['res5']", "MODEL.ROI_HEADS.IN_FEATURES", "['res5']", "MODEL.RESNETS.RES5_DILATION", "2"])
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
R = 2000
model.backbone.use_plan = os.environ.get("INFER_PLAN", "1") == "1"
model.roi_heads.scores_only = os.environ.get("INFER_SCORES_ONLY", "0") == "1"  # what a TTA pass runs (no per-pass NMS)
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("INFER_SIZES", "224x224,480x640,688x920,1200x1600").split(",")]
for (H, W) in SIZES:
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (3, H, W), generator=g).float().cuda()
    x0 = torch.rand(R, generator=g) * (W - 60)
    y0 = torch.rand(R, generator=g) * (H - 60)
    bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20)
    bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20)
    p = Instances((H, W))
    p.proposal_boxes = Boxes(torch.stack([x0, y0, x0 + bw, y0 + bh], 1).cuda())
    p.objectness_logits = torch.rand(R, generator=g).cuda()
    inp = [{"image": img, "proposals": p, "height": H, "width": W}]
    for _ in range(2):
        model.inference(inp, do_postprocess=False)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 10
    for _ in range(n):
        model.inference(inp, do_postprocess=False)
    torch.cuda.synchronize()
    print("%4dx%-4d: %.2f ms per inference pass (R=%d)" % (H, W, (time.time() - t0) / n * 1e3, R), flush=True)
