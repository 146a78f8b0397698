#!/usr/bin/env python
"""Per-(kernel, grid size) launch statistics of a rocprofv3 results .db (one kernel name often covers very different layer
shapes):  python tools/prof_by_grid.py <results.db> <name substring> [...]"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    kv = "kernels" if "kernels" in views else [v for v in views if "kernel" in v.lower()][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kv)]
    gcol = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid")][0]
    for pat in sys.argv[2:]:
        q = ("select name, %s, count(*), avg(end - start), min(end - start) from %s where name like ? group by name, %s order by name, %s"
             % (gcol, kv, gcol, gcol))
        for name, grid, n, avg, mn in cur.execute(q, ("%" + pat + "%",)):
            print("%-60s grid %8d  calls %5d  avg %8.2f us  min %8.2f us" % (name[:60], grid, n, avg / 1e3, mn / 1e3))


if __name__ == "__main__":
    main()
