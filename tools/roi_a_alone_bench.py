import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"
for knob in (0, 1):
    ops.tune(14, knob)
    for (H, W, R) in ((14, 14, 2000), (43, 58, 1800), (50, 76, 2000), (63, 92, 1947), (75, 122, 1500)):
        C = 1024
        feat = torch.randn((1, H, W, C), device=dev).to(torch.bfloat16)
        g = torch.Generator().manual_seed(0)
        IW, IH = W * 16, H * 16
        x0, y0 = torch.rand(R, generator=g) * (IW - 40), torch.rand(R, generator=g) * (IH - 40)
        bw, bh = 20 + torch.rand(R, generator=g) * (IW - x0 - 20), 20 + torch.rand(R, generator=g) * (IH - y0 - 20)
        rois = torch.stack([torch.zeros(R), x0, y0, (x0 + bw).clamp(max=IW), (y0 + bh).clamp(max=IH)], 1).to(dev)
        obj = torch.rand(R, device=dev)
        A = torch.zeros((R, C * 49), dtype=torch.bfloat16, device=dev)
        f = lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A)
        for _ in range(3): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        print("A alone, knob14=%d, map %3d x %3d: %.1f us" % (knob, H, W, a.elapsed_time(b) / 10 * 1e3))
