"""Temporal vs SPATIAL sharing of the chip between the fc6 weight-gradient GEMM (MFMA-bound) and the optimizer pass
(HBM-bound).  Round 3 measured that the two on the same CUs cost about the SUM of their stand-alone times; this probe
asks whether the interference is CU-local (then disjoint CU sets fix it) or lives in the memory system (then nothing on
the CU side does):

  1. what a CU mask means on this device (hipExtStreamCreateWithCUMask): which (XCC, SE, CU) a masked stream's workgroups
     land on, for the low 8*(32-k) bits and for the complement
  2. the optimizer pass on k CUs per XCD (k = 2 .. 32): bytes/s per CU
  3. the dW GEMM (6 exact rounds: [2048 x 49152] over K = 2000, TN form) on 8*(32-k) resident workgroups, alone
  4. both at once on disjoint CU sets, against both at once on shared CUs and against one after the other

  python tools/cu_partition_probe.py [k,k,...]"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
H = ctypes.CDLL(os.path.join(ROOT, "tools", "build", "libcu_mask_helper.so"))
H.cum_stream_create.restype = ctypes.c_void_p
H.cum_stream_create.argtypes = [ctypes.c_void_p, ctypes.c_int]
H.cum_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
TUNE_GEMM_NWG = 18
dev = "cuda"
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = H.cum_stream_create(words, len(words))
    if not h:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed")
    return torch.cuda.ExternalStream(h)


def where(stream, blocks=4096, lds=0):
    out = torch.zeros((blocks, 2), dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        H.cum_where(out.data_ptr(), blocks, 64, lds, 2000, stream.cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    hw, xcc = o[:, 0], o[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = {x: len({i for i in ids if i[0] == x}) for x in sorted({i[0] for i in ids})}
    return len(ids), per_xcc


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 6, 8, 12]
    D1, K1, R = 2048, 50176, 2048
    NM = 49152
    torch.manual_seed(0)
    dPT = (torch.randn((D1, R), device=dev) * 0.05).to(torch.bfloat16)
    A = (torch.randn((2000, K1), device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn((D1 * K1,), device=dev) * 0.02
    mom = torch.randn_like(w) * 0.01
    sh = torch.zeros((D1 * K1,), dtype=torch.bfloat16, device=dev)
    g16 = torch.zeros((D1, K1), dtype=torch.bfloat16, device=dev)
    seg = np.zeros(1, dtype=[("off", "<i8"), ("cnt", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
    seg[0] = (0, D1 * K1, 0.0, 5e-4)
    seg_dev = torch.from_numpy(seg.view(np.uint8)).to(dev)
    nbytes = 20.0 * D1 * NM
    gflop = 2.0 * D1 * NM * 2000

    def gemm():
        ops.gemm_tn(dPT, A[:, :NM], D1, NM, R, 2000, out=g16[:, :NM].unsqueeze(0))

    def sgd():
        ops.sgd_step_block(w, mom, g16.view(-1), seg_dev, 0, D1, 0, NM, K1, 0.9, False, shadow=sh, grad_off=0)

    print("device: %d CUs" % NCU)
    # ---- 1. mask semantics
    full = torch.cuda.Stream()
    print("unmasked stream: %d distinct CUs, per XCC %s" % where(full))
    for lo, hi in ((0, 208), (208, 256), (0, 32), (0, 8)):
        try:
            s = masked_stream(range(lo, hi))
            print("mask bits %3d..%3d: %3d distinct CUs, per XCC %s" % ((lo, hi) + where(s)))
        except Exception as ex:  # noqa: BLE001
            print("mask bits %d..%d: %r" % (lo, hi, ex))
    # ---- 2-4
    t_g = timeit(gemm)
    ops.tune(ops.TUNE_SGD_GRID, 512)
    t_s = timeit(sgd)
    print("\nGEMM alone, all CUs            : %7.1f us (%.0f TFLOP/s)" % (t_g, gflop / t_g / 1e6))
    print("SGD alone, all CUs, grid 512   : %7.1f us (%.2f TB/s)" % (t_s, nbytes / t_s / 1e6))
    s2 = torch.cuda.Stream()

    def both(sa, sb):
        def run():
            main_s = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main_s)
            sa.wait_event(ev)
            sb.wait_event(ev)
            with torch.cuda.stream(sa):
                gemm()
            with torch.cuda.stream(sb):
                sgd()
            main_s.wait_stream(sa)
            main_s.wait_stream(sb)
        return run

    def each(sa, sb, n=6):
        """own duration of each kernel while the other runs (events on its stream)"""
        tg, ts = [], []
        for _ in range(n):
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            with torch.cuda.stream(sa):
                e[0].record(sa)
                gemm()
                e[1].record(sa)
            with torch.cuda.stream(sb):
                e[2].record(sb)
                sgd()
                e[3].record(sb)
            torch.cuda.synchronize()
            tg.append(e[0].elapsed_time(e[1]) * 1e3)
            ts.append(e[2].elapsed_time(e[3]) * 1e3)
        return sum(tg[1:]) / (n - 1), sum(ts[1:]) / (n - 1)

    t_b = timeit(both(full, s2))
    eg, es = each(full, s2)
    print("both, SHARED CUs (two streams) : %7.1f us   (GEMM %6.1f, SGD %6.1f; sum alone %6.1f)" % (t_b, eg, es, t_g + t_s))
    for k in ks:
        n_g = 8 * (32 - k)
        try:
            sg = masked_stream(range(0, n_g))
            so = masked_stream(range(n_g, 256))
        except Exception as ex:  # noqa: BLE001
            print("k=%d: %r" % (k, ex))
            continue
        old = ops.tune(TUNE_GEMM_NWG, n_g)
        for grid in (8 * k * 4, 8 * k * 8):
            ops.tune(ops.TUNE_SGD_GRID, grid)
            with torch.cuda.stream(so):
                t_sk = timeit(sgd, n=5)
            with torch.cuda.stream(sg):
                t_gk = timeit(gemm, n=5)
            t_bk = timeit(both(sg, so), n=6)
            eg, es = each(sg, so)
            print("k=%2d (%3d CUs GEMM | %3d CUs SGD grid %4d): GEMM alone %6.1f  SGD alone %6.1f (%.2f TB/s, %.1f GB/s/CU)  "
                  "both %6.1f (GEMM %6.1f, SGD %6.1f)" % (k, n_g, 8 * k, grid, t_gk, t_sk, nbytes / t_sk / 1e6,
                                                          nbytes / t_sk / 1e3 / (8 * k), t_bk, eg, es))
        # the GEMM on fewer workgroups but NO masks: do the SGD workgroups find the free CUs by themselves?
        ops.tune(ops.TUNE_SGD_GRID, 8 * k * 8)
        t_bu = timeit(both(full, s2), n=6)
        eg, es = each(full, s2)
        print("      same grids, unmasked streams: both %6.1f (GEMM %6.1f, SGD %6.1f)" % (t_bu, eg, es))
        ops.tune(TUNE_GEMM_NWG, old)
    ops.tune(ops.TUNE_SGD_GRID, 512)


if __name__ == "__main__":
    main()
