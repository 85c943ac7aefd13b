"""One inference pass (scores only, what a TTA pass runs) of the chosen workload at one image size, a few times, for
`rocprofv3 --kernel-trace --stats`: kernel-time split of a pass.  usage: WORKLOAD=r50c4|r50dc5 python tools/infer_trace.py H W [R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from __graft_entry__ import load_package

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.modeling import build_model
from drn_wsod_pytorch_amd.structures import Boxes, Instances

H, W = int(sys.argv[1]), int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
cfg = bench.build_cfg(pkg, "cuda")
if os.environ.get("WORKLOAD", "r50c4") == "r50dc5":
    cfg.merge_from_list(["MODEL.RESNETS.OUT_FEATURES", "['res5']", "MODEL.ROI_HEADS.IN_FEATURES", "['res5']", "MODEL.RESNETS.RES5_DILATION", "2"])
for kv in filter(None, os.environ.get("TUNE", "").split(",")):  # TUNE=22=32,...: drn_tune knobs for A/B runs
    from drn_wsod_pytorch_amd import ops as _ops
    _ops.tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
model.roi_heads.scores_only = True
g = torch.Generator().manual_seed(1)
img = torch.randint(0, 256, (3, H, W), generator=g).float().cuda()
x0 = torch.rand(R, generator=g) * (W - 60)
y0 = torch.rand(R, generator=g) * (H - 60)
bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20) * float(os.environ.get("BOX_FRAC", "1.0"))
bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20) * float(os.environ.get("BOX_FRAC", "1.0"))
p = Instances((H, W))
p.proposal_boxes = Boxes(torch.stack([x0, y0, x0 + bw, y0 + bh], 1).cuda())
p.objectness_logits = torch.rand(R, generator=g).cuda()
inp = [{"image": img, "proposals": p, "height": H, "width": W}]
for _ in range(12):
    model.inference(inp, do_postprocess=False)
torch.cuda.synchronize()
