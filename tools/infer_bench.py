"""Inference (eval) pass of R50-C4 at TTA-like image sizes: wall time per image and the top kernels."""
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
model = build_model(cfg)
bench.init_weights(model, seed=0)
model.eval()
R = 2000
for (H, W) in [(224, 224), (480, 640), (688, 920), (1200, 1600)]:
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
