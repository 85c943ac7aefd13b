"""What does a cross-stream wait cost, and whom?  Stream A runs a chain of N kernels (K1 K2 K1 K2 ...); after every K1 an event is
recorded on A.  Variants: (a) nothing else; (b) the event is recorded, nobody waits; (c) stream B waits for each event and runs a
tiny kernel; (d) like (c), and A then waits for B's event before its next K1 (a join, as the step's main stream does).
Wall time per pair of A's kernels, by HIP events on A around 200 pairs.
  python tools/stream_wait_probe.py"""
import torch

dev = "cuda"
x = torch.randn(64 << 20, device=dev)           # K1: 256 MB read + write (~100 us)
y = torch.randn(8 << 20, device=dev)            # K2: ~15 us
z = torch.zeros(1024, device=dev)
A, B = torch.cuda.Stream(), torch.cuda.Stream()


def run(variant, n=200):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(A):
        for it in range(n + 20):
            if it == 20:
                t0.record(A)
            x.mul_(1.0001)
            if variant != "a":
                e = torch.cuda.Event()
                e.record(A)
                if variant in ("c", "d"):
                    B.wait_event(e)
                    with torch.cuda.stream(B):
                        z.add_(1.0)
                        if variant == "d":
                            eb = torch.cuda.Event()
                            eb.record(B)
            y.mul_(1.0001)
            if variant == "d":
                A.wait_event(eb)
        t1.record(A)
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n * 1e3


for rep in range(3):
    print("  ".join("%s: %7.1f us" % (v, run(v)) for v in ("a", "b", "c", "d")), flush=True)
