# kernel table of one replayed bf16 224^2 training step (scratch: gpurun_out/exp_step/$1.txt)
R=$(pwd); OUT=$R/gpurun_out/exp_step; mkdir -p $OUT; tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp; cd $R
STEP="--steps 3 --warmup 3 --skip-cpu --skip-roofline --skip-secondary --no-settle --dtype bf16 --imsize 224"
rm -rf $OUT/raw_$tag
rocprofv3 --kernel-trace --stats -d $OUT/raw_$tag -- python bench.py $STEP > $OUT/$tag.stdout 2> $OUT/$tag.err || true
db=$(find $OUT/raw_$tag -name "*results.db" | head -1)
python tools/prof_summary.py $db laststep > $OUT/$tag.txt
rm -rf $OUT/raw_$tag
grep -n "wgrad" $OUT/$tag.txt | head -20
head -12 $OUT/$tag.txt
tail -1 $OUT/$tag.stdout | cut -c1-220
python bench.py --steps 20 --warmup 5 --skip-cpu --skip-roofline --skip-secondary --dtype bf16 --imsize 224 | cut -c1-220
