"""cProfile of the host side of one full TTA call (tools/tta_bench.py's R50-C4 case): where the wall time of the call goes
beyond the 16 device passes.  TTA_WORKLOAD=r50c4|r50dc5."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from __graft_entry__ import load_package

pkg = load_package()
pkg.set_precision("bf16")
from drn_wsod_pytorch_amd.modeling import build_model
from drn_wsod_pytorch_amd.modeling.tta import GeneralizedRCNNWithTTAAVG
from drn_wsod_pytorch_amd.structures import Boxes, Instances

H, W, R = 375, 500, 2000
wl = os.environ.get("TTA_WORKLOAD", "r50c4")
cfg = bench.build_cfg(pkg, "cuda")
extra = ["TEST.AUG.ENABLED", "True", "TEST.AUG.MIN_SIZES", "(480, 576, 672, 768, 864, 960, 1056, 1152)", "TEST.AUG.MAX_SIZE", "4000",
         "TEST.AUG.FLIP", "True"]
if wl == "r50dc5":
    extra += ["MODEL.RESNETS.OUT_FEATURES", "['res5']", "MODEL.ROI_HEADS.IN_FEATURES", "['res5']", "MODEL.RESNETS.RES5_DILATION", "2",
              "MODEL.ROI_BOX_HEAD.DAN_DIM", "[2048, 4096]"]
cfg.merge_from_list(extra)
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
g = torch.Generator().manual_seed(1)
img = torch.randint(0, 256, (3, H, W), generator=g).to(torch.uint8)
x0 = torch.rand(R, generator=g) * (W - 60)
y0 = torch.rand(R, generator=g) * (H - 60)
bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20) * 0.6
bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20) * 0.6
p = Instances((H, W))
p.proposal_boxes = Boxes(torch.stack([x0, y0, x0 + bw, y0 + bh], 1))
p.objectness_logits = torch.rand(R, generator=g)
inp = {"image": img, "proposals": p, "height": H, "width": W}
tta = GeneralizedRCNNWithTTAAVG(cfg, model)
for _ in range(3):
    tta([inp])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    tta([inp])
torch.cuda.synchronize()
print("%s: %.1f ms per TTA call (unprofiled)" % (wl, (time.perf_counter() - t0) / 3 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tta([inp])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
