"""Launch the roofline kernel (fc6 forward GEMM of BASELINE configs[1]) a few times on its own, for the PMC passes:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out1 -- python tools/pmc_gemm.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out2 -- python tools/pmc_gemm.py
(separate passes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 - MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from __graft_entry__ import load_package

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
M, N, K, S = [int(x) for x in os.environ.get("PMC_SHAPE", "2000,2048,50176,4").split(",")]  # default: fc6 forward
A = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
B = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty((S, M, N), dtype=torch.bfloat16 if os.environ.get("PMC_BF16_OUT") else torch.float32, device="cuda")
TN = os.environ.get("PMC_TN")  # the second operand K-major (drn_gemm_tn): Bt [kb_rows][N], kb_rows = K - 48 like R = 2000
if TN:
    kb = K - 48
    Bt = (torch.randn((kb, N), device="cuda") * 0.05).to(torch.bfloat16)
SGDP = os.environ.get("PMC_SGDP")  # round 4: the fused fc6 dW + SGD launch (drn_gemm_tn_sgd): PMC_SHAPE = D1,K1,Rpad,1
if SGDP:
    import numpy as np

    kb = K - 48
    Bt = (torch.randn((kb, N), device="cuda") * 0.05).to(torch.bfloat16)
    w = torch.randn((M, N), device="cuda") * 0.02
    mom = torch.randn((M, N), device="cuda") * 0.01
    sh = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    g16 = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (0, M * N, 0.0, 5e-4)
    seg_dev = torch.from_numpy(seg.view(np.uint8)).to("cuda")
    for _ in range(6):
        assert ops.gemm_tn_sgd(A, Bt, M, N, K, kb, g16, w, mom, sh, seg_dev, 0.9, False, 1.0)
    torch.cuda.synchronize()
    print("done")
    sys.exit(0)
for _ in range(6):
    if TN:
        ops.gemm_tn(A, Bt, M, N, K, kb, out=out, splits=S)
    else:
        ops.gemm_nt(A, B, M, N, K, out=out, splits=S)
torch.cuda.synchronize()
print("done")
