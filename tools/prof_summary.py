#!/usr/bin/env python
"""Summarise a rocprofv3 run (sqlite `*_results.db` or `*_kernel_stats.csv`) into a compact per-kernel table
(the files committed under profiles/).  usage: tools/prof_summary.py <db-or-csv> [steps] > profiles/<name>.txt"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("at::native::", "")
    return name[:100]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = []
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            rows.append((short(name), int(calls), float(total) / 1e3, float(avg), float(pct)))  # db durations are in us
    else:
        for r in csv.DictReader(open(path)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                         float(r["Percentage"])))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path.split("/")[-1])
    print("# total GPU kernel time %.3f ms%s" % (tot, (" = %.3f ms/step over %g steps" % (tot / steps, steps)) if steps else ""))
    print("%-100s %7s %11s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for n, c, t, a, p in rows:
        if p < 0.05:
            continue
        print("%-100s %7d %11.3f %10.2f %6.2f" % (n, c, t, a, p))


if __name__ == "__main__":
    main()
