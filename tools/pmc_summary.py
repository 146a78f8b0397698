#!/usr/bin/env python
"""Condense rocprofv3 `--pmc <COUNTER>` counter_collection.csv files into the per-(kernel, grid) table committed under profiles/:
median counter value per launch shape, in bytes for FETCH_SIZE / WRITE_SIZE (the counters are in KiB).
usage: tools/pmc_summary.py <name substring> <csv> [<csv> ...]      (one csv per counter pass)"""
import csv
import statistics
import sys


def main():
    pat, paths = sys.argv[1], sys.argv[2:]
    table = {}
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                if pat not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"][:90], int(r["Grid_Size"]))
                table.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    counters = sorted({c for v in table.values() for c in v})
    print("%-90s %10s %6s " % ("kernel", "grid", "calls") + " ".join("%16s" % c for c in counters))
    tot = {c: 0.0 for c in counters}
    for (name, grid), v in sorted(table.items()):
        n = max(len(x) for x in v.values())
        cells = []
        for c in counters:
            med = statistics.median(v[c]) if c in v else float("nan")
            scale = 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 1.0
            tot[c] += med * scale
            cells.append("%16.0f" % (med * scale))
        print("%-90s %10d %6d " % (name, grid, n) + " ".join(cells))
    print("%-90s %10s %6s " % ("sum over launch shapes (median per shape)", "", "") + " ".join("%16.0f" % tot[c] for c in counters))
    if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
        print("# HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half the fetched bytes, profiles/r02_fetch_calibration.txt) "
              "= %.1f MB" % ((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1e6))


if __name__ == "__main__":
    main()
