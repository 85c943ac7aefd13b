"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd .db) into a per-kernel table (text)."""
import glob
import sqlite3
import sys


def main(path, out=None):
    dbs = glob.glob(path + "/**/*.db", recursive=True)
    assert dbs, "no rocpd database under " + path
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (%s)" % dbs[0].split("/")[-1],
             "# total kernel time %.1f us over %d dispatches" % (tot, sum(r[1] for r in rows)),
             "%-96s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for r in rows:
        lines.append("%-96s %7d %12.1f %10.2f %10.2f %10.2f %5.1f%%" % (r[0][:96], r[1], r[2], r[3], r[4], r[5],
                                                                       100 * r[2] / tot))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
