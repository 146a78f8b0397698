#!/bin/bash
# Regenerates the rocprofv3 summaries committed under profiles/ for one round (run on the GPU box: `gpurun -- bash tools/make_profiles.sh r03`).
# Everything is written under gpurun_out/<tag>/ ; copy the *.txt / *.json you want judged into profiles/.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
STEP="--steps 3 --warmup 3 --skip-cpu --skip-roofline --skip-secondary --no-settle"
prof() {   # name, then the command; the command's stdout (bench.py's JSON line) is kept next to the profile: SAME process
  local name=$1; shift
  rm -rf $OUT/raw_$name
  rocprofv3 --kernel-trace --stats -d $OUT/raw_$name -- "$@" > $OUT/${name}.stdout 2> $OUT/${name}.err
  find $OUT/raw_$name -name "*results.db" | head -1
}
# one replayed training step, fp32 256^2 and bf16 224^2
db=$(prof step_fp32 python bench.py $STEP);                  python tools/prof_summary.py $db laststep > $OUT/bench_fp32_step.txt
db=$(prof step_bf16 python bench.py $STEP --dtype bf16 --imsize 224); python tools/prof_summary.py $db laststep > $OUT/bench_bf16_224_step.txt
# the roofline leg: the launches `roofline.achieved` is computed from, one row per (kernel, grid)
db=$(prof roof_fp32 python bench.py --roofline-only --product-only)
python tools/prof_by_grid.py $db conv3x3_direct_group_kernel "conv3x3_direct_kernel<" > $OUT/roofline_leg_fp32.txt
# (the HIP-event figures of the SAME profiled process, so that the two clocks can be compared line by line)
python - $OUT/roof_fp32.stdout >> $OUT/roofline_leg_fp32.txt <<'PYEOF'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("# HIP events in the same (profiled) process: grouped launch %.4f ms per timestep; single launches %s; hoisted convs %.4f ms per iteration"
      % (r["ms_per_timestep"], [(x["HxW"], x["ms_product"]) for x in r["per_scale"]], r["hoisted_convs_ms_per_iteration"]))
PYEOF
db=$(prof roof_bf16 python bench.py --roofline-only --product-only --dtype bf16 --imsize 224)
python tools/prof_by_grid.py $db "conv_blk_dec_group_kernel" > $OUT/roofline_leg_bf16_224.txt
python - $OUT/roof_bf16.stdout >> $OUT/roofline_leg_bf16_224.txt <<'PYEOF'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("# HIP events in the same (profiled) process: grouped launch %.4f ms per timestep; single launches %s; hoisted convs %.4f ms per iteration"
      % (r["ms_per_timestep"], [(x["HxW"], x["ms_product"]) for x in r["per_scale"]], r["hoisted_convs_ms_per_iteration"]))
PYEOF
# HBM counters of the gate launches (separate passes per counter: they do not fit one), both dtypes
for dt in fp32 bf16; do
  sz=256; [ $dt = bf16 ] && sz=224
  csvs=""
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/raw_pmc_${dt}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/raw_pmc_${dt}_$c -o pmc -- python bench.py --roofline-only --product-only --kernel-iters 4 --dtype $dt --imsize $sz > /dev/null 2>&1
    csvs="$csvs $(find $OUT/raw_pmc_${dt}_$c -name '*counter_collection.csv' | head -1)"
  done
  pat="conv3x3_direct_group_kernel"; [ $dt = bf16 ] && pat="conv_blk_dec_group_kernel"
  python tools/pmc_summary.py "$pat" $csvs > $OUT/gate_pmc_$dt.txt
done
# per-kernel HBM traffic of one EAGER bf16 224^2 step (blocked bf16 trunk), two PMC passes joined by tools/step_traffic.py
csvs=""
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/raw_st_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/raw_st_$c -o pmc -- python bench.py --no-graph --steps 2 --warmup 2 --skip-cpu --skip-roofline --skip-secondary --no-settle --dtype bf16 --imsize 224 > /dev/null 2>&1
  csvs="$csvs $(find $OUT/raw_st_$c -name '*counter_collection.csv' | head -1)"
done
python tools/step_traffic.py $csvs > $OUT/step_traffic_bf16_224.txt
# ---- round 5: BASELINE configs[4]'s per-GPU workload (512x1024, T=20, batch 8, bf16): one replayed step, the roofline leg, its PMC bytes
CFG4="--dtype bf16 --imsize 512 --imsize-w 1024 --batch 8 --T 20"
db=$(prof step_cfg4 python bench.py $STEP $CFG4); python tools/prof_summary.py $db laststep > $OUT/bench_bf16_cfg4_step.txt
db=$(prof roof_cfg4 python bench.py --roofline-only --product-only $CFG4)
python tools/prof_by_grid.py $db "conv_blk_dec_group_kernel" > $OUT/roofline_leg_bf16_cfg4.txt
python - $OUT/roof_cfg4.stdout >> $OUT/roofline_leg_bf16_cfg4.txt <<'PYEOF'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("# HIP events in the same (profiled) process: grouped launch %.4f ms per timestep; single launches %s; hoisted convs %.4f ms per iteration"
      % (r["ms_per_timestep"], [(x["HxW"], x["ms_product"]) for x in r["per_scale"]], r["hoisted_convs_ms_per_iteration"]))
PYEOF
csvs=""
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/raw_pmc_cfg4_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/raw_pmc_cfg4_$c -o pmc -- python bench.py --roofline-only --product-only --kernel-iters 4 $CFG4 > /dev/null 2>&1
  csvs="$csvs $(find $OUT/raw_pmc_cfg4_$c -name '*counter_collection.csv' | head -1)"
done
python tools/pmc_summary.py "conv_blk_dec_group_kernel" $csvs > $OUT/gate_pmc_bf16_cfg4.txt
# ---- round 5: MFMA-busy counters (their own pass, kernel-trace only) for the fp32 gate launch and the fp32 1x1 GEMM
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
rm -rf $OUT/raw_mfma_gate
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/raw_mfma_gate -o pmc -- python bench.py --roofline-only --product-only --kernel-iters 4 > /dev/null 2>&1
python tools/pmc_summary.py "conv3x3_direct" $(find $OUT/raw_mfma_gate -name '*counter_collection.csv' | head -1) > $OUT/gate_mfma_busy_fp32.txt
rm -rf $OUT/raw_mfma_gemm
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/raw_mfma_gemm -o pmc -- python tools/exp/gemm1x1_sweep.py --tiles 0 --iters 3 > /dev/null 2>&1
python tools/pmc_summary.py "conv_igemm_kernel" $(find $OUT/raw_mfma_gemm -name '*counter_collection.csv' | head -1) > $OUT/gemm1x1_mfma_busy_fp32.txt
# ---- round 6: inference (test() as a replayed graph) fp32 256^2 and bf16 224^2: one step each; the Winograd kernel: per-shape A/B table and
# its MFMA-busy counters (own pass)
INF="--inference --steps 3 --warmup 4 --skip-roofline"
db=$(prof inf_fp32 python bench.py $INF);                          python tools/prof_summary.py $db lastperiod > $OUT/inference_fp32_step.txt
db=$(prof inf_bf16 python bench.py $INF --dtype bf16 --imsize 224); python tools/prof_summary.py $db lastperiod > $OUT/inference_bf16_224_step.txt
python tools/wino_bench.py --iters 30 > $OUT/wino_per_shape.txt 2>&1
rm -rf $OUT/raw_mfma_wino
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/raw_mfma_wino -o pmc -- python tools/wino_bench.py --iters 3 --only-wino > /dev/null 2>&1
python tools/pmc_summary.py "conv_wino_f32_kernel" $(find $OUT/raw_mfma_wino -name '*counter_collection.csv' | head -1) > $OUT/wino_mfma_busy_fp32.txt
rm -rf $OUT/raw_lds_wino
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d $OUT/raw_lds_wino -o pmc -- python tools/wino_bench.py --iters 3 --only-wino > /dev/null 2>&1
python tools/pmc_summary.py "conv_wino_f32_kernel" $(find $OUT/raw_lds_wino -name '*counter_collection.csv' | head -1) > $OUT/wino_lds_counters_fp32.txt
# blocked bf16 conv against the fp32-storage bf16 conv on the trunk's shapes
rm -rf $OUT/raw_*
ls -la $OUT
