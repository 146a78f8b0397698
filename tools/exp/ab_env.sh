# A/B of one environment variable on one box: bash tools/exp/ab_env.sh VAR "v1 v2 ..." "<bench flags>"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in $2; do
  env $1=$v python bench.py $3 --skip-roofline --skip-cpu --skip-secondary --steps 30 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v: %.3f ms' % r['ms_per_step'])"
done; done
