# kernel times of the 3x3 weight-gradient group kernels in a bf16 step, per library variant (rsis_amd/lib/exp/librsis_*.so)
R=$(pwd); OUT=$R/gpurun_out/exp_step; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $R
STEP="--steps 3 --warmup 3 --skip-cpu --skip-roofline --skip-secondary --no-settle --dtype bf16 --imsize 224"
for v in base "$@"; do
  if [ $v = base ]; then unset RSIS_HIP_LIB; else export RSIS_HIP_LIB=$R/rsis_amd/lib/exp/librsis_$v.so; fi
  rm -rf $OUT/raw
  rocprofv3 --kernel-trace --stats -d $OUT/raw -- python bench.py $STEP > $OUT/v_$v.stdout 2> $OUT/v_$v.err || tail -3 $OUT/v_$v.err
  db=$(find $OUT/raw -name "*results.db" | head -1)
  python tools/prof_summary.py $db laststep > $OUT/v_$v.txt
  echo "== $v: $(grep 'total GPU kernel time' $OUT/v_$v.txt)"
  grep "wgrad3_tr\|wgrad1_bf16" $OUT/v_$v.txt | awk '{printf "   %-60s %s %s\n", $1" "$2" "$3" "$4" "$5, $(NF-3), $(NF-2)}'
done
rm -rf $OUT/raw
