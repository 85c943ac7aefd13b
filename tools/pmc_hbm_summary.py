"""Summarise the two PMC passes of tools/pmc_hbm.py.  usage: pmc_hbm_summary.py <fetch_dir> <write_dir> <out.json>
Counters are in KiB.  Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 for 16-B/lane streaming reads
(the optimizer's reads; the pooling kernel reads 0.4 MB, irrelevant); WRITE_SIZE is uncalibrated in general - here both
kernels have an exactly known written byte count, so the record doubles as its calibration for 16-B streaming stores."""
import glob
import json
import sqlite3
import sys


def rows(d, kernel, name):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    q = "select value, duration from counters_collection where kernel_name like ? and counter_name=?"
    return list(cur.execute(q, ("%" + kernel + "%", name)))[1:]  # first launch: cold


R, K1 = 2000, 1024 * 49
n = 1024 * K1
spec = {
    "roi_pool7_lane_kernel": {"read": 14 * 14 * 1024 * 2 + R * 5 * 4 + R * 4, "write": R * K1 * 2,
                               "what": "A [2000 x 50176] bf16 written as 98-byte runs (2-byte stores), 0.4 MB map + boxes read"},
    "sgd_kernel": {"read": n * (4 + 4 + 2), "write": n * (4 + 4 + 2),
                   "what": "per parameter: w, momentum fp32 + bf16 gradient read; w, momentum fp32 + bf16 shadow written"},
}
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python tools/pmc_hbm.py (separate passes)",
       "corrections": "KiB->B x1024; FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE x1", "kernels": {}}
for k, s in spec.items():
    f, w = rows(sys.argv[1], k, "FETCH_SIZE"), rows(sys.argv[2], k, "WRITE_SIZE")
    fetch = sum(v for v, _ in f) / len(f) * 1024 * 2
    write = sum(v for v, _ in w) / len(w) * 1024
    dur = sum(d for _, d in f) / len(f) / 1e3
    alg = s["read"] + s["write"]
    out["kernels"][k] = {"launches": len(f), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                         "traffic_bytes_per_launch": fetch + write, "algorithmic_read_bytes": s["read"],
                         "algorithmic_write_bytes": s["write"], "traffic_over_algorithmic": (fetch + write) / alg,
                         "avg_duration_us_under_pmc": dur, "algorithmic_GBps_under_pmc": alg / dur / 1e3, "bytes": s["what"]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
