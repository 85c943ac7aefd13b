"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db) into a per-kernel table (text).  Launches of one kernel
with different grids are different shapes of the workload (the 256x256 GEMM serves the fc6 forward, the fc6 dW slabs and
fc7), so rows are split by (kernel, grid): `wgs` = workgroups per launch as x*y."""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_db import main_db


def main(path, out=None):
    dbs = [main_db(path)]
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select name, grid_x / max(workgroup_x, 1), grid_y / max(workgroup_y, 1), count(*), "
                       "sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                       "group by name, grid_x, grid_y, workgroup_x, workgroup_y order by 5 desc").fetchall()
    tot = sum(r[4] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % dbs[0].split("/")[-1],
             "# total kernel time %.1f us over %d dispatches; rows split by (kernel, grid)" % (tot, sum(r[3] for r in rows)),
             "%-84s %11s %7s %12s %10s %10s %10s %6s" % ("kernel", "wgs", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for r in rows:
        if r[4] < 0.0005 * tot:
            continue
        lines.append("%-84s %11s %7d %12.1f %10.2f %10.2f %10.2f %5.1f%%" % (
            r[0].replace("(anonymous namespace)::", "").replace("void ", "")[:84], "%dx%d" % (r[1], r[2]), r[3], r[4], r[5],
            r[6], r[7], 100 * r[4] / tot))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
