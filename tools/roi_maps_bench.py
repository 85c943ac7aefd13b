import importlib, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"
for kv in sys.argv[1:]:  # tune knobs: 15=76 (LDS budget), 4=1024 (threads), 10=4 (chunks per block) ...
    ops.tune(*[int(x) for x in kv.split("=")])
TAIL = os.environ.get("ROI_TAIL", "1") == "1"  # round-3 step: A + the A^T rows of channels >= 1000 (0: all of A^T)
ops.ROI_WORKSPACE = os.environ.get("ROI_WS", "1") == "1"  # 0: without the chunk-major scratch copy (drn_roi_pool_nhwc_ws)
ALONE = os.environ.get("ROI_A_ALONE", "0") == "1"  # A alone (no A^T): what the training step (round 4 on) and inference launch
for (H, W, R) in ((14, 14, 2000), (36, 50, 1800), (43, 58, 1800), (50, 76, 2000), (63, 92, 1947), (75, 122, 1500)):
    C = 1024
    feat = torch.randn((1, H, W, C), device=dev).to(torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    IW, IH = W * 16, H * 16
    x0, y0 = torch.rand(R, generator=g) * (IW - 40), torch.rand(R, generator=g) * (IH - 40)
    bw, bh = 20 + torch.rand(R, generator=g) * (IW - x0 - 20), 20 + torch.rand(R, generator=g) * (IH - y0 - 20)
    rois = torch.stack([torch.zeros(R), x0, y0, (x0 + bw).clamp(max=IW), (y0 + bh).clamp(max=IH)], 1).to(dev)
    if os.environ.get("ROI_SORT", "0") == "1":  # (experiment) similar-sized ROIs next to each other: no load imbalance between the waves of a block
        rois = rois[torch.argsort((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))].contiguous()
    obj = torch.rand(R, device=dev)
    K = C * 49
    A = torch.zeros((R, K), dtype=torch.bfloat16, device=dev)
    AT = torch.zeros((K, ops.kpad(R, torch.bfloat16)), dtype=torch.bfloat16, device=dev)
    f = (lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A)) if ALONE else \
        (lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1 / 16, out=A, out_t=AT, t_first_channel=1000 if TAIL else 0))
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    win = float(((rois[:, 3] - rois[:, 1]) / 16 * (rois[:, 4] - rois[:, 2]) / 16).mean())
    print("map %3d x %3d, R = %4d, mean window %.0f px: %.1f us" % (H, W, R, win, a.elapsed_time(b) / 10 * 1e3))
