"""cProfile of the host side of plain inference passes (tools/infer_bench.py's loop at one size): python tools/infer_host_profile.py H W"""
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
from drn_wsod_pytorch_amd.structures import Boxes, Instances

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
cfg = bench.build_cfg(pkg, "cuda")
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
R = 2000
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
for _ in range(3):
    model.inference(inp, do_postprocess=False)
torch.cuda.synchronize()
for so in (False, True):
    model.roi_heads.scores_only = so
    t0 = time.perf_counter()
    for _ in range(20):
        model.inference(inp, do_postprocess=False)
    torch.cuda.synchronize()
    print("%dx%d scores_only=%s: %.3f ms per pass" % (H, W, so, (time.perf_counter() - t0) / 20 * 1e3))
model.roi_heads.scores_only = False
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    model.inference(inp, do_postprocess=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:6000])
