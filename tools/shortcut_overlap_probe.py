"""Would the projection shortcut of a bottleneck's first block gain from running BESIDE the block's conv1 / conv2 (a second
stream) instead of in front of them?  The first block of res5 and of res4 of the dilated-C5 recipe at 800 x 1216 (99 x 151 cells):
shortcut 1x1 Cin -> 4c | conv1 1x1 Cin -> c, conv2 3x3 c -> c (dil 2) | conv3 1x1 c -> 4c + shortcut.  Sequential on one stream
against fork / join on two (events), 50 repetitions.
  python tools/shortcut_overlap_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev, dt = "cuda", torch.bfloat16
H, W = 99, 151


def mk(cout, cin, k):
    kk = cin * k * k
    return (torch.randn((cout, ops.kpad(kk, dt)), device=dev) * (1.0 / kk ** 0.5)).to(dt), torch.ones(cout, device=dev), torch.zeros(cout, device=dev)


for name, cin, c in (("res5 block 1", 1024, 512), ("res4 block 1", 512, 256)):
    x = (torch.randn((1, H, W, cin), device=dev) * 0.5).to(dt)
    ws, ss, bs = mk(4 * c, cin, 1)
    w1, s1, b1 = mk(c, cin, 1)
    w2, s2, b2 = mk(c, c, 3)
    w3, s3, b3 = mk(4 * c, c, 1)
    side = torch.cuda.Stream()

    def seq():
        sc = ops.conv2d_nhwc(x, ws, 4 * c, 1, 1, scale=ss, bias=bs)
        y = ops.conv2d_nhwc(x, w1, c, 1, 1, scale=s1, bias=b1, relu=True)
        y = ops.conv2d_nhwc(y, w2, c, 3, 3, pad=2, dil=2, scale=s2, bias=b2, relu=True)
        return ops.conv2d_nhwc(y, w3, 4 * c, 1, 1, scale=s3, bias=b3, residual=sc, relu=True)

    def fork():
        main = torch.cuda.current_stream()
        e0 = torch.cuda.Event()
        e0.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e0)
            sc = ops.conv2d_nhwc(x, ws, 4 * c, 1, 1, scale=ss, bias=bs)
            e1 = torch.cuda.Event()
            e1.record(side)
        y = ops.conv2d_nhwc(x, w1, c, 1, 1, scale=s1, bias=b1, relu=True)
        y = ops.conv2d_nhwc(y, w2, c, 3, 3, pad=2, dil=2, scale=s2, bias=b2, relu=True)
        main.wait_event(e1)
        return ops.conv2d_nhwc(y, w3, 4 * c, 1, 1, scale=s3, bias=b3, residual=sc, relu=True)

    assert torch.equal(seq(), fork())
    for fn, label in ((seq, "one stream"), (fork, "shortcut on a second stream"), (seq, "one stream"), (fork, "shortcut on a second stream")):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        print("%s  %-28s %7.1f us per block" % (name, label, a.elapsed_time(b) / 50 * 1e3), flush=True)
