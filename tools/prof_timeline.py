"""Print the kernel timeline of one steady-state step from a rocprofv3 --kernel-trace database.
usage: prof_timeline.py <dir> [number_of_steps]"""
import glob
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import main_db

db = main_db(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, queue_id, grid_x, grid_y, workgroup_x from kernels order by start"))
# step boundary: the roi pool kernel of the main stream runs once per step
marks = [i for i, r in enumerate(rows) if "roi_pool7" in r[0] or "roi_kernel" in r[0]]
mid = len(marks) // 2  # a steady-state step in the middle of the run (the end of a bench run holds the stand-alone
lo, hi = marks[mid], marks[mid + back]  # HBM-roofline launches); `back` = number of consecutive steps to print
t0 = rows[lo][1]
for name, s, e, q, gx, gy, wx in rows[lo:hi + 1]:
    short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", name)[:58]
    print("%9.1f %8.1f  q%-3d %-58s g=%dx%d" % ((s - t0) / 1e3, (e - s) / 1e3, q, short, gx // max(wx, 1), gy))
