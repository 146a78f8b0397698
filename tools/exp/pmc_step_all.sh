# SQ counters of every kernel of one bf16 training step (scratch): where do waves spend their lifetime?
R=$(pwd); OUT=$R/gpurun_out/pmc_all; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for grp in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/raw -o pmc -- python bench.py --no-graph --steps 2 --warmup 2 --skip-cpu --skip-roofline --skip-secondary --no-settle --dtype ${DT:-bf16} --imsize ${SZ:-224} > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  cp $f $OUT/pass$i.csv; i=$((i+1))
done
rm -rf $OUT/raw
python - $OUT/pass0.csv $OUT/pass1.csv $OUT/pass2.csv <<'PY'
import csv, sys, collections
tab = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in sys.argv[1:]:
    seen = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:70]
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_WAVES"):
            seen[k] += 1
    for k, v in seen.items():
        cnt[k] = max(cnt[k], v)
rows = sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
print("%-70s %6s %10s %6s %6s %6s %6s %9s %6s" % ("kernel (all launches of the profiled steps summed)", "calls", "waveMcyc", "wait%", "stall%", "issue%", "valu%", "valu/wave", "mfma%"))
for k, v in rows[:40]:
    wc = v.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-70s %6d %10.1f %6.1f %6.1f %6.1f %6.1f %9.0f %6.1f" % (k, cnt[k], wc * 4 / 1e6, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
          100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1), 100 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc)))
PY
