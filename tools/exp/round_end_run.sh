set -x
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_fp32 -o run -- python $R/bench.py --steps 3 --warmup 1 --skip-cpu --skip-roofline --skip-secondary --no-settle > $R/gpurun_out/final/prof_fp32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_bf16 -o run -- python $R/bench.py --dtype bf16 --imsize 224 --steps 3 --warmup 1 --skip-cpu --skip-roofline --skip-secondary --no-settle > $R/gpurun_out/final/prof_bf16.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof_roof -o run -- python $R/bench.py --roofline-only > $R/gpurun_out/final/prof_roof.log 2>&1
cd $R
cat gpurun_out/final/pytest_gpu.txt
cut -c1-300 gpurun_out/final/bench_default.json
