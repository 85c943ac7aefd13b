"""GPU busy fraction from a rocprofv3 --kernel-trace database: union of all kernel intervals (any queue) over the window between
the pooling kernels of steady-state steps, plus the idle gaps longer than a threshold with the kernels around them.
usage: prof_busy.py <dir> [gap_us=20]"""
import glob
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import main_db

db = main_db(sys.argv[1])
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, queue_id from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "roi_pool7" in r[0] or "roi_kernel" in r[0]]
lo, hi = marks[len(marks) // 4], marks[3 * len(marks) // 4]
steps = 3 * len(marks) // 4 - len(marks) // 4
win = rows[lo:hi]
t0, t1 = win[0][1], max(r[2] for r in win)
busy, cur_s, cur_e = 0, win[0][1], win[0][2]
gaps = []
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n)[:50]
last_name = win[0][0]
for name, s, e, q in win[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if (s - cur_e) / 1e3 >= gap_us:
            gaps.append(((s - cur_e) / 1e3, short(last_name), short(name)))
        cur_s, cur_e, last_name = s, e, name
    elif e > cur_e:
        cur_e, last_name = e, name
busy += cur_e - cur_s
print("window %.1f ms, %d steps: %.3f ms per step; GPU busy (union of kernels) %.1f %%; %d idle gaps >= %.0f us (%.1f us per step)"
      % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps, 100.0 * busy / (t1 - t0), len(gaps), gap_us, sum(g[0] for g in gaps) / steps))
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    agg[(a, b)][0] += 1
    agg[(a, b)][1] += g
for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print("  %4d x avg %6.1f us idle between  %-50s -> %s" % (n, t / n, a, b))
