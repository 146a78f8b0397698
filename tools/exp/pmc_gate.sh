R=$(pwd); OUT=$R/gpurun_out/pmc_gate; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
for grp in "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" "SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/raw -o pmc -- python bench.py --roofline-only --product-only --kernel-iters 4 --dtype bf16 --imsize 224 > /dev/null 2> $OUT/err.txt
  f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "conv_blk_dec_group_kernel<1" $f | cut -c1-30,91-200 || tail -3 $OUT/err.txt
done
rm -rf $OUT/raw
