# A/B of two builds of the library on one box: `bash tools/exp/ab_lib.sh <other.so>` (relative to the repo root); the product build is ""
cd $GRAFT_REPO_ROOT
OTHER=$1
run() { # label, extra bench flags
  for lib in "$OTHER" "" "$OTHER" ""; do
    RSIS_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py $2 --skip-roofline --skip-cpu --skip-secondary --steps 30 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 lib=[$lib]: %.3f ms' % r['ms_per_step'])"
  done
}
run cfg4 "--dtype bf16 --imsize 512 --imsize-w 1024 --batch 8 --T 20"
run bf16_224 "--dtype bf16 --imsize 224"
run fp32_256 ""
