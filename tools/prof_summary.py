#!/usr/bin/env python
"""Summarise a rocprofv3 run (sqlite `*_results.db` or `*_kernel_stats.csv`) into a compact per-kernel table
(the files committed under profiles/).  usage: tools/prof_summary.py <db-or-csv> [steps | laststep] > profiles/<name>.txt
`laststep` (sqlite only): the table of ONE training step -- the kernels between the last two optimizer updates (two adam_kernel
launches end a step), whatever warm-up / capture steps the run contained."""
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
    last = len(sys.argv) > 2 and sys.argv[2] in ("laststep", "lastperiod")
    steps = float(sys.argv[2]) if len(sys.argv) > 2 and not last else None
    rows = []
    if last:
        cur = sqlite3.connect(path).cursor()
        ks = list(cur.execute("select name,start,end from kernels order by start"))
        if sys.argv[2] == "lastperiod":
            # `lastperiod` (a run that ends with replays of ONE captured graph and has no optimizer step to delimit it: bench.py --inference):
            # the shortest L >= 50 with names[-L:] == names[-2L:-L] -- the kernels of the last replay
            # (a few kernels may follow the last replay -- the caller's finiteness check: up to 40 trailing launches are skipped)
            names, seg = [k[0] for k in ks], None
            for tail in range(0, 41):
                nm = names[:len(names) - tail]
                L = next((L for L in range(50, len(nm) // 2) if nm[-L:] == nm[-2 * L:-L]), None)
                if L is not None:
                    seg = ks[len(nm) - L:len(nm)]
                    break
            if seg is None:
                raise SystemExit("no repeating launch sequence found at the end of the trace")
        else:
            ad = [i for i, k in enumerate(ks) if "adam_kernel" in k[0]]
            seg = ks[ad[-4] + 1: ad[-2] + 1]
        agg = {}
        for n, s0, e0 in seg:
            a = agg.setdefault(short(n), [0, 0.0])
            a[0] += 1
            a[1] += (e0 - s0) / 1e6
        tot = sum(a[1] for a in agg.values())
        rows = [(n, a[0], a[1], 1e3 * a[1] / a[0], 100.0 * a[1] / tot) for n, a in agg.items()]
        print("# one %s: %d kernel launches, first start to last end %.3f ms" % ("replay of the captured graph" if sys.argv[2] == "lastperiod" else "training step", len(seg), (seg[-1][2] - seg[0][1]) / 1e6))
    elif path.endswith(".db"):
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
