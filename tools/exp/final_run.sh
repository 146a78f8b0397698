set -x
bash tools/make_profiles.sh r04 > gpurun_out/make_profiles.log 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.err
