"""Runs of consecutive dispatches of the same (kernel, grid) in a rocprofv3 --kernel-trace database: count, mean kernel
duration and mean start-to-start period (= duration + the gap to the next launch).  usage: prof_runs.py <dir> [name-filter]"""
import glob
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import main_db

db = main_db(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
runs = []
for name, s, e, gx, wx in rows:
    key = (name, gx // max(wx, 1))
    if runs and runs[-1][0] == key:
        runs[-1][1].append((s, e))
    else:
        runs.append((key, [(s, e)]))
print("%-64s %6s %6s %9s %9s %9s" % ("kernel", "wgs", "calls", "avg_us", "min_us", "period_us"))
for (name, wgs), se in runs:
    if flt not in name or len(se) < 5:
        continue
    se = se[len(se) // 3:]  # steady part of the run
    d = [(e - s) / 1e3 for s, e in se]
    per = [(se[i + 1][0] - se[i][0]) / 1e3 for i in range(len(se) - 1)]
    short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", name)[:64]
    print("%-64s %6d %6d %9.2f %9.2f %9.2f" % (short, wgs, len(se), sum(d) / len(d), min(d), sum(per) / max(len(per), 1)))
