# SQ counters of the 3x3 blk weight-gradient kernel on one decoder shape (scratch)
R=$(pwd); OUT=$R/gpurun_out/pmc_wgrad; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
SHAPE=${1:-320x192->256}
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/raw -o pmc -- python tools/exp/wgrad_blk_bench.py "$SHAPE" > /dev/null 2> $OUT/err_$tag.txt
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py wgrad3_tr $f | cut -c1-40,91-200 || tail -3 $OUT/err_$tag.txt
done
rm -rf $OUT/raw
