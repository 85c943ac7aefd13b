"""One ROIPool shape, A alone, for PMC / trace passes: [ROI_C=1024 ROI_STRIDE=16] python tools/roi_one.py H W R [launches] [knob=value ...]"""
import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
H, W, R = [int(x) for x in sys.argv[1:4]]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
for kv in sys.argv[5:]:
    ops.tune(*[int(x) for x in kv.split("=")])
C, dev = int(os.environ.get("ROI_C", 1024)), "cuda"
S = int(os.environ.get("ROI_STRIDE", 16))
feat = torch.randn((1, H, W, C), device=dev).to(torch.bfloat16)
g = torch.Generator().manual_seed(0)
IW, IH = W * S, H * S
x0, y0 = torch.rand(R, generator=g) * (IW - 40), torch.rand(R, generator=g) * (IH - 40)
bw, bh = 20 + torch.rand(R, generator=g) * (IW - x0 - 20), 20 + torch.rand(R, generator=g) * (IH - y0 - 20)
rois = torch.stack([torch.zeros(R), x0, y0, (x0 + bw).clamp(max=IW), (y0 + bh).clamp(max=IH)], 1).to(dev)
obj = torch.rand(R, device=dev)
A = torch.zeros((R, C * 49), dtype=torch.bfloat16, device=dev)
for _ in range(n):
    ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / S, out=A)
torch.cuda.synchronize()
print("done")
