"""Summarise the two PMC passes of tools/pmc_gemm.py into profiles/<name>.json.
usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
16-B/lane streaming reads (the GEMM's only read pattern) -> x2; WRITE_SIZE taken as is (it matches the split-K
partial-buffer size to 1%)."""
import glob
import json
import sqlite3
import sys


def rows(d, name):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    q = "select value, duration from counters_collection where kernel_name like '%gemm_nt%' and counter_name=?"
    return list(cur.execute(q, (name,)))


f = rows(sys.argv[1], "FETCH_SIZE")[1:]
w = rows(sys.argv[2], "WRITE_SIZE")[1:]
fetch = sum(v for v, _ in f) / len(f) * 1024 * 2
write = sum(v for v, _ in w) / len(w) * 1024
import os

M, N, K, S = [int(x) for x in os.environ.get("PMC_SHAPE", "2000,2048,50176,4").split(",")]  # default: fc6 forward
alg = (M + N) * K * 2 + S * M * N * (2 if os.environ.get("PMC_BF16_OUT") else 4)
out = {"kernel": os.environ.get("PMC_KERNEL", "gemm_nt256_kernel<bf16,PIPE> fc6 fwd"), "shape": [M, N, K], "splits": S, "launches": len(f),
       "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
       "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch + write) / alg,
       "avg_duration_us_under_pmc": sum(d for _, d in f) / len(f) / 1e3,
       "corrections": "KiB->B x1024; FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE x1",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/pmc_gemm.py (separate passes)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
