import torch, sys
sys.path.insert(0, "/root/repo")
n = 401849408
for dt, nm in ((torch.uint8, "fill u8"),):
    t = torch.empty(n, dtype=dt, device="cuda")
    src = torch.empty(n, dtype=dt, device="cuda")
    for name, fn, byt in (("zero_ (write only)", lambda: t.zero_(), n), ("copy_ (read + write)", lambda: t.copy_(src), 2 * n)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print("%-24s %d MB: %.1f us  %.2f TB/s" % (name, n / 1e6, us, byt / us / 1e6))
