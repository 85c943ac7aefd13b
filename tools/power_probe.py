"""Socket power and shader clock under the step's two regimes (hwmon sysfs, sampled every ~10 ms while a loop of launches
keeps the GPU busy for ~2 s each):
   fc6 dW GEMM loop (MFMA-bound)   |   optimizer pass loop (HBM-bound)   |   both at once on two streams   |
   the GEMM on half the CUs (CU-masked stream)   |   idle
Answers: is the GEMM running at the package power cap (then an HBM-bound kernel beside it takes its power, and
'overlap' can only buy the difference in energy, not the whole stand-alone time of the hidden kernel)?

  python tools/power_probe.py"""
import glob
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("drn_wsod_pytorch_amd.ops")
dev = "cuda"


def find_hwmon():
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            f = {}
            for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    f[name] = p
            if f:
                out.append((card, f))
    return out


def rd(p):
    try:
        with open(p) as fh:
            return float(fh.read().strip())
    except Exception:  # noqa: BLE001
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, files, dt=0.01):
        super().__init__(daemon=True)
        self.files, self.dt, self.rows, self.stop = files, dt, [], False

    def run(self):
        while not self.stop:
            self.rows.append((time.perf_counter(),) + tuple(rd(p) for p in self.files.values()))
            time.sleep(self.dt)


def main():
    hw = find_hwmon()
    if not hw:
        print("no hwmon files under /sys/class/drm/card*/device/hwmon")
        return
    # a box may expose several cards in sysfs while one GPU is visible to HIP: take the card whose PCI address is ours
    card, files = hw[0]
    try:
        pr = torch.cuda.get_device_properties(0)
        want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for c, f in hw:
            if want in os.path.realpath(c):
                card, files = c, f
        print("HIP device 0 is PCI %s" % want)
    except Exception as ex:  # noqa: BLE001
        print("could not match the PCI address (%r): first card" % (ex,))
    print("sampling %s -> %s: %s" % (card, os.path.realpath(card), ", ".join(files)))
    print("cards with hwmon: %s" % ", ".join(os.path.realpath(c).rsplit("/", 1)[-1] for c, _ in hw))
    if "power1_cap" in files:
        print("power cap: %.0f W" % (rd(files["power1_cap"]) / 1e6))
    D1, K1, R, NM = 2048, 50176, 2048, 49152
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
    Wt = sh.view(D1, K1)
    part = torch.empty((4, 2000, D1), dtype=torch.float32, device=dev)
    s2 = torch.cuda.Stream()

    def gemm():
        ops.gemm_tn(dPT, A[:, :NM], D1, NM, R, 2000, out=g16[:, :NM].unsqueeze(0))

    def fwd():
        ops.gemm_nt(A, Wt, 2000, D1, K1, out=part, splits=4)

    def sgd():
        ops.sgd_step_block(w, mom, g16.view(-1), seg_dev, 0, D1, 0, NM, K1, 0.9, False, shadow=sh, grad_off=0)

    def both():
        gemm()
        with torch.cuda.stream(s2):
            sgd()

    def idle():
        time.sleep(0.02)

    half = None
    try:
        import ctypes

        H = ctypes.CDLL(os.path.join(ROOT, "tools", "build", "libcu_mask_helper.so"))
        H.cum_stream_create.restype = ctypes.c_void_p
        H.cum_stream_create.argtypes = [ctypes.c_void_p, ctypes.c_int]
        words = (ctypes.c_uint32 * 8)()
        for b in range(128):
            words[b // 32] |= 1 << (b % 32)
        half = torch.cuda.ExternalStream(H.cum_stream_create(words, 8))
    except Exception as ex:  # noqa: BLE001
        print("no CU-masked stream: %r" % (ex,))

    def gemm_half():
        with torch.cuda.stream(half):
            gemm()

    Wr = (torch.randn((D1, K1), device=dev) * 0.02).to(torch.bfloat16)
    Az = torch.zeros_like(A)

    def fwd_rand():
        ops.gemm_nt(A, Wr, 2000, D1, K1, out=part, splits=4)

    def fwd_zero():
        ops.gemm_nt(Az, Wt, 2000, D1, K1, out=part, splits=4)

    cases = [("idle", idle, None), ("fc6 forward GEMM, random A and W", fwd_rand, 2.0 * 2000 * D1 * K1),
             ("fc6 forward GEMM, random A, W = 0", fwd, 2.0 * 2000 * D1 * K1),
             ("fc6 forward GEMM, A = 0 and W = 0", fwd_zero, 2.0 * 2000 * D1 * K1), ("fc6 dW GEMM (TN, 6 rounds)", gemm, 2.0 * D1 * NM * 2000),              ("optimizer pass (20 B/param)", sgd, None), ("dW GEMM + optimizer pass, two streams", both, 2.0 * D1 * NM * 2000)]
    if half is not None:
        cases.append(("dW GEMM on 128 CUs (masked stream, 128 workgroups)", gemm_half, 2.0 * D1 * NM * 2000))
    names = list(files)
    print("%-52s %9s %9s  %s" % ("case", "launch us", "TFLOP/s", "  ".join("%s(mean/max)" % n for n in names)))
    for name, fn, flop in cases:
        if fn is gemm_half:
            ops.tune(18, 128)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        smp = Sampler(files)
        smp.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(20):
                fn()
            n += 20
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        smp.stop = True
        smp.join()
        if fn is gemm_half:
            ops.tune(18, 0)
        rows = np.array([r[1:] for r in smp.rows if r[0] - t0 > 0.5])  # skip the ramp
        cols = []
        for i, nme in enumerate(names):
            v = rows[:, i]
            sc = 1e-6 if nme.startswith("power") else (1e-6 if nme.startswith("freq") else 1e-3)
            cols.append("%8.0f /%8.0f" % (np.nanmean(v) * sc, np.nanmax(v) * sc))
        us = dt / n * 1e6
        print("%-52s %9.1f %9s  %s" % (name, us, "%.0f" % (flop / us / 1e6) if flop else "-", "  ".join(cols)))
    print("(power in W, freq in MHz, temp in C; the first 0.5 s of every case is dropped)")


if __name__ == "__main__":
    main()
