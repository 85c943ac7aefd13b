"""What can a WRITE-ONLY kernel reach on this GPU?  ROIPool writes 401 MB and reads 0.4 MB; the fused SGD pass (half
reads, half writes) reaches 6.4-6.8 TB/s.  Times, with HIP events over 20 launches each, on 401-MB and 1-GB bf16 buffers:
fill_ (16-byte stores from an elementwise kernel), zero_ (memset), copy_ (read + write) and the ROIPool launch itself."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for nbytes in (401408000, 1 << 30):
    x = torch.empty((nbytes // 2,), dtype=torch.bfloat16, device=dev)
    y = torch.randn((nbytes // 2,), device=dev).to(torch.bfloat16)
    for name, fn, moved in (("fill_(1.0)  write-only", lambda: x.fill_(1.0), nbytes),
                            ("zero_()     write-only", lambda: x.zero_(), nbytes),
                            ("copy_       read+write", lambda: x.copy_(y), 2 * nbytes)):
        t = timed(fn)
        print("%8.1f MB  %-24s %7.1f us  %6.2f TB/s" % (nbytes / 1e6, name, t * 1e6, moved / t / 1e12))

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
t = timed(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / 16, out=A, out_t=AT))
print("%8.1f MB  %-24s %7.1f us  %6.2f TB/s" % (2 * R * K1 * 2 / 1e6, "ROIPool A + A^T", t * 1e6, 2 * R * K1 * 2 / t / 1e12))
t = timed(lambda: ops.roi_pool_nhwc(feat, rois, obj, 7, 1.0 / 16, out=A))
print("%8.1f MB  %-24s %7.1f us  %6.2f TB/s" % (R * K1 * 2 / 1e6, "ROIPool A only", t * 1e6, R * K1 * 2 / t / 1e12))
