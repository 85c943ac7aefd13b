"""Launch the two HBM-bound kernels of the step (bench.py's `roofline_hbm` entries) on their own, for PMC passes:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out1 -- python tools/pmc_hbm.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out2 -- python tools/pmc_hbm.py
(separate passes; tools/pmc_hbm_round.sh runs both and summarises them).
  * roi_pool7_lane_kernel (round 4): 14x14x1024 bf16 map, 2000 SURVEY 8(d) boxes -> A [2000 x 50176] (no A^T row is needed any more)
  * sgd_kernel<shadow, bf16 grad>: one fc6 row slab (1024 x 50176 parameters) of RANDOM fp32 weights / momentum, a bf16
    gradient bucket and the bf16 shadow (zero-filled operands clock higher: MI355X_MICROARCH.md, DVFS note)"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"
g = torch.Generator().manual_seed(7)
R, C, H, W = 2000, 1024, 14, 14
feat = torch.rand((1, H, W, C), generator=g).to(dev).to(torch.bfloat16)
x0, y0 = torch.rand(R, generator=g) * 184, torch.rand(R, generator=g) * 184
bw, bh = 20 + torch.rand(R, generator=g) * (224 - x0 - 20), 20 + torch.rand(R, generator=g) * (224 - y0 - 20)
rois = torch.stack([torch.zeros(R), x0, y0, (x0 + bw).clamp(max=224), (y0 + bh).clamp(max=224)], 1).to(dev)
obj = torch.sort(torch.rand(R, generator=g), descending=True).values.to(dev)
K1 = C * 49
A = torch.zeros((R, ops.kpad(K1, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
AT = torch.zeros((K1, ops.kpad(R, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
for _ in range(6):
    ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / 16, out=A, out_t=AT, t_first_channel=C)  # what the step launches since round 4: A alone, lane-per-bin kernel
torch.cuda.synchronize()

n = 1024 * K1
w = torch.randn((n,), device=dev) * 0.01
mom = torch.randn((n,), device=dev) * 0.001
grad = (torch.randn((n,), device=dev) * 0.001).to(torch.bfloat16)
shadow = w.to(torch.bfloat16)
seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
seg[0] = (0, n, 1e-3, 5e-4)
seg_dev = torch.from_numpy(seg.view(np.uint8)).to(dev)
for _ in range(6):
    ops.sgd_step(w, mom, grad, seg_dev, 1, 0.9, False, shadow=shadow, grad_off=0)
torch.cuda.synchronize()
print("done")
