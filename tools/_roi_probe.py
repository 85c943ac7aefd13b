import importlib, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev="cuda"
def timed(fn, n=20):
    # replayed from a hipGraph: the Python call costs ~50 us, more than the knocked-out kernels take
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
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
print("DBG=%s  A+AT %.1f us" % (os.environ.get("DRN_ROI_DBG", "0"), t))
