R=$(pwd); OUT=$R/gpurun_out/pmc_upconv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
python tools/exp/upconv_bench.py
for grp in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/raw -o pmc -- python tools/exp/upconv_bench.py > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py upconv $f | cut -c1-40,91-200 || tail -3 $OUT/err.txt
done
rm -rf $OUT/raw
