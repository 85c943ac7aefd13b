"""Time the PCL device path (drn_pcl_adjacency + drn_pcl_refine) on SURVEY 8(d)-shaped synthetic proposals and report
the sizes that drive it (top-cluster members per labelled class, centres)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
from drn_wsod_pytorch_amd import ops  # noqa: E402


def main():
    R = int(os.environ.get("PCL_R", 2000))
    K, NB, G = 20, 3, int(os.environ.get("PCL_G", 2))
    mode = os.environ.get("PCL_BOXES", "random")
    g = torch.Generator().manual_seed(7)
    W = H = 224.0
    if mode == "random":
        x0 = torch.rand(R, generator=g) * (W - 40)
        y0 = torch.rand(R, generator=g) * (H - 40)
        bw = 20 + torch.rand(R, generator=g) * (W - x0 - 20)
        bh = 20 + torch.rand(R, generator=g) * (H - y0 - 20)
    else:  # clustered around 12 objects
        c = torch.randint(0, 12, (R,), generator=g)
        cx0, cy0 = torch.rand(12, generator=g) * 120, torch.rand(12, generator=g) * 120
        cw, ch = 40 + torch.rand(12, generator=g) * 60, 40 + torch.rand(12, generator=g) * 60
        j = torch.randn(R, 4, generator=g) * 5
        x0, y0 = (cx0[c] + j[:, 0]).clamp(0, W - 25), (cy0[c] + j[:, 1]).clamp(0, H - 25)
        bw, bh = (cw[c] + j[:, 2]).clamp(min=20), (ch[c] + j[:, 3]).clamp(min=20)
    boxes = torch.stack([x0, y0, (x0 + bw).clamp(max=W), (y0 + bh).clamp(max=H)], 1).contiguous().cuda()
    a = torch.randn(R, K, generator=g) * 2
    b = torch.randn(R, K, generator=g) * 3
    ws = (torch.softmax(a, 1) * torch.softmax(b, 0)).contiguous().cuda()
    logits = (torch.randn(R, NB * (K + 1), generator=g) * 2).cuda()
    onehot = torch.zeros(K)
    onehot[torch.randperm(K, generator=g)[:G]] = 1
    onehot = onehot.cuda()
    dl = torch.zeros_like(logits)
    cols = [b_ * (K + 1) for b_ in range(NB)]

    def run():
        adj = ops.pcl_adjacency(boxes, 0.4)
        return ops.pcl_refine(logits, cols, K, ws, boxes, adj, onehot, dl)

    out = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    print("R=%d boxes=%s G=%d: %.1f us per call (adjacency + softmax + refine, %d branches); centres per branch %s; "
          "fg rows per branch %s" % (R, mode, G, e0.elapsed_time(e1) * 1e3 / n, NB, [int(o["n_pc"]) for o in out],
                                     [int((o["labels"] > 0).sum()) for o in out]))


    if os.environ.get("PCL_PROFILE"):  # library built with -DPCL_PROFILE: phase clocks of branch 0 (100 MHz ticks)
        t = out[-1]["pc_scores"][-10:].cpu().numpy()
        names = ["gather+sort", "prefix+cuts", "k-means search", "members+degrees", "greedy loop", "top-5", "assign+stats",
                 "loss+dlogits"]
        print("  last branch phases (us):", {n_: round(float(v) / 100.0, 1) for n_, v in zip(names, t[:8])},
              "greedy iterations", int(t[8]))


if __name__ == "__main__":
    main()
