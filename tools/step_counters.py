#!/usr/bin/env python
"""Where the kernels of one training step wait: per-kernel sums of SQ counters over the LAST step of
`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES SQ_WAVES --output-format csv -- python bench.py --no-graph --steps 2 --warmup 2 ...`.
SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked on s_waitcnt / s_barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY, in quad-cycles
(MI355X_MICROARCH.md, rocprofv3 PMC slots); printed as fractions of the wave cycles.  usage: tools/step_counters.py counters.csv"""
import csv
import re
import sys

NAMES = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES",
         "SQ_WAVES"]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("at::native::", "")[:70]


def main():
    rows = {}
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t": (int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)) / 1e3})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = sorted(rows)
    ad = [i for i, k in enumerate(order) if "adam_kernel" in rows[k]["name"]]
    order = order[ad[-4] + 1: ad[-2] + 1]
    agg = {}
    for k in order:
        d = rows[k]
        a = agg.setdefault(short(d["name"]), {"n": 0, "t": 0.0})
        a["n"] += 1
        a["t"] += d["t"]
        for c in NAMES:
            a[c] = a.get(c, 0.0) + d.get(c, 0.0)
    tot = sum(a["t"] for a in agg.values())
    print("# last training step: %d launches, %.2f ms of kernel time under the counter collection" % (sum(a["n"] for a in agg.values()), tot / 1e3))
    print("# fractions of SQ_WAVE_CYCLES: parked = SQ_WAIT_ANY (s_waitcnt / s_barrier), stall = SQ_WAIT_INST_ANY (issue stalls; lds = the part on LDS),")
    print("# active = SQ_ACTIVE_INST_ANY; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES): MFMA-pipe occupancy, a RELATIVE figure (the two")
    print("# counters are summed over different numbers of instances: compare kernels, do not read it as a fraction)")
    print("%-70s %5s %9s %7s %7s %7s %7s %7s %9s" % ("kernel", "calls", "total_us", "parked", "stall", "(lds)", "active", "mfma", "waves/call"))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
        if a["t"] < 0.004 * tot:
            continue
        wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        print("%-70s %5d %9.1f %7.2f %7.2f %7.2f %7.2f %7.2f %9.0f" % (
            n, a["n"], a["t"], a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc,
            a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4.0 * a.get("SQ_BUSY_CYCLES", 0), 1.0), a.get("SQ_WAVES", 0) / a["n"]))


if __name__ == "__main__":
    main()
