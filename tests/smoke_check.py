"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run():
    from oracle import wsod_oracle as O

    ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    # ROIPool -> fc GEMM -> bias/ReLU on a small problem, fp32 parity mode
    feat = torch.from_numpy(rs.standard_normal((1, 64, 14, 14)).astype(np.float32))
    x0, y0 = rs.rand(96) * 150, rs.rand(96) * 150
    rois = torch.from_numpy(np.stack([np.zeros(96), x0, y0, x0 + 20 + rs.rand(96) * 50, y0 + 20 + rs.rand(96) * 50], 1)
                            .astype(np.float32))
    obj = torch.rand(96)
    W = torch.from_numpy(rs.standard_normal((128, 64 * 49)).astype(np.float32) * 0.02)
    b = torch.from_numpy(rs.standard_normal(128).astype(np.float32) * 0.1)
    pooled, _ = O.roi_pool_forward(feat, rois, 7, 1 / 16)
    ref = torch.relu((pooled * (obj + 1).view(-1, 1, 1, 1)).flatten(1) @ W.t() + b)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(dev)
    A = ops.roi_pool_nhwc(fd, rois.to(dev), obj.to(dev), 7, 1 / 16)
    P = ops.gemm_nt(A, W.to(dev), 96, 128, 64 * 49, splits=2)
    out = torch.zeros((96, 128), device=dev)
    ops.bias_act_fwd(P, 96, 128, b.to(dev), True, out=out)
    torch.cuda.synchronize()
    err = float((out.cpu() - ref).abs().max())
    assert err < 1e-3, err
    print("smoke ok: max abs err vs oracle %.3e" % err)
