# SQ / TCC counters of the weight-gradient group kernels inside one bf16 training step (scratch)
R=$(pwd); OUT=$R/gpurun_out/pmc_wgrad; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
for grp in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"; do
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/raw -o pmc -- python bench.py --no-graph --steps 2 --warmup 2 --skip-cpu --skip-roofline --skip-secondary --no-settle --dtype bf16 --imsize 224 > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py wgrad $f | cut -c1-50,91-200 || tail -3 $OUT/err.txt
done
rm -rf $OUT/raw
