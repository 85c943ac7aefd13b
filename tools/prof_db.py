"""Pick the rocpd database of the traced MAIN process: a run that starts a child (bench.py's side measurements) leaves one
database per process, and the first one the file system lists may be the child's.  The main process is the one with the
most kernel dispatches."""
import glob
import sqlite3


def main_db(path):
    dbs = glob.glob(path + "/**/*.db", recursive=True)
    assert dbs, "no rocpd database under " + path
    if len(dbs) == 1:
        return dbs[0]
    best, n_best = dbs[0], -1
    for d in dbs:
        try:
            n = sqlite3.connect(d).execute("select count(*) from kernels").fetchone()[0]
        except sqlite3.Error:
            n = -1
        if n > n_best:
            best, n_best = d, n
    return best
