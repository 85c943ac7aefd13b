"""Where the hand-over gaps of the step come from: kernel trace + HIP API trace of the same run (rocprofv3 --kernel-trace
--hip-trace), one steady-state step, kernels of the main queue and the host's API calls merged in time - is the launch that
follows a gap ENQUEUED late (host-bound) or dispatched late (device-side)?
usage: gap_probe.py <dir> [schema]"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import main_db

db = main_db(sys.argv[1])
con = sqlite3.connect(db)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view') order by name")]
if len(sys.argv) > 2:
    for n in names:
        try:
            cols = [c[1] for c in cur.execute("pragma table_info('%s')" % n)]
            cnt = cur.execute("select count(*) from '%s'" % n).fetchone()[0]
            print(n, cnt, cols)
        except sqlite3.Error as e:
            print(n, "?", e)
    sys.exit(0)
rows = list(cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "roi_pool7" in r[0]]
mid = len(marks) // 2
lo, hi = marks[mid], marks[mid + 1]
t0, t1 = rows[lo][1], rows[hi][2]
ev = []
for name, s, e, q, gx, wx in rows[lo:hi + 1]:
    short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", name)[:50]
    ev.append((s, "K q%-2d %-50s dur %7.1f" % (q, short, (e - s) / 1e3)))
api = None
for cand in ("regions", "regions_and_samples", "rocpd_region", "api"):
    if cand in names:
        api = cand
        break
if api:
    cols = [c[1] for c in cur.execute("pragma table_info('%s')" % api)]
    ncol = "name" if "name" in cols else cols[0]
    try:
        for name, s, e in cur.execute("select %s, start, end from '%s' where start >= ? and start <= ? order by start" % (ncol, api), (t0 - 2000000, t1)):
            if "Launch" in str(name) or "Graph" in str(name) or "Event" in str(name) or "Wait" in str(name) or "Memcpy" in str(name):
                ev.append((s, "   host %-40s dur %7.1f" % (str(name)[:40], (e - s) / 1e3)))
    except sqlite3.Error as ex:
        print("api table", api, "unreadable:", ex, cols)
ev.sort()
for s, txt in ev:
    print("%10.1f  %s" % ((s - t0) / 1e3, txt))
