R=$(pwd); OUT=$R/gpurun_out/exp_inf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
rm -rf $OUT/raw
rocprofv3 --kernel-trace --stats -d $OUT/raw -- python tools/bench_inference.py --graph ${INF_ARGS:---dtype bf16} > $OUT/out.txt 2> $OUT/err.txt
db=$(find $OUT/raw -name "*results.db" | head -1)
python tools/prof_summary.py $db | head -45
rm -rf $OUT/raw
tail -1 $OUT/out.txt
