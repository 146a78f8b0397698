#!/usr/bin/env python
"""Per-kernel HBM traffic of one training step: joins the FETCH_SIZE and WRITE_SIZE passes of
`rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python bench.py --no-graph --steps 2 --warmup 2 ...` (one pass per counter)
by kernel name and prints, per kernel, launches / time / bytes (2 x FETCH_SIZE + WRITE_SIZE) / achieved GB/s for the LAST step
(the dispatches between the last two pairs of adam_kernel launches).  usage: tools/step_traffic.py fetch.csv write.csv"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("at::native::", "")[:86]


def load(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0,
                             int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0)))
    rows.sort()
    ad = [i for i, r in enumerate(rows) if "adam_kernel" in r[1]]
    return rows[ad[-4] + 1: ad[-2] + 1]


def main():
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    agg = {}
    for (_d, n, v, s, e) in fe:
        a = agg.setdefault(short(n), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        a[2] += 2.0 * v
    for (_d, n, v, _s, _e) in wr:
        agg.setdefault(short(n), [0, 0.0, 0.0, 0.0])[3] += v
    tot_t = sum(a[1] for a in agg.values())
    tot_b = sum(a[2] + a[3] for a in agg.values())
    print("# last training step: %d launches, %.2f ms of kernel time (serialised by the counter collection), %.2f GB of HBM traffic" %
          (sum(a[0] for a in agg.values()), tot_t / 1e3, tot_b / 1e9))
    print("%-86s %6s %10s %10s %10s %9s" % ("kernel", "calls", "total_us", "read_MB", "write_MB", "GB/s"))
    for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3])):
        if a[2] + a[3] < 0.002 * tot_b:
            continue
        print("%-86s %6d %10.1f %10.1f %10.1f %9.0f" % (n, a[0], a[1], a[2] / 1e6, a[3] / 1e6, (a[2] + a[3]) / max(a[1], 1e-9) / 1e3))


if __name__ == "__main__":
    main()
