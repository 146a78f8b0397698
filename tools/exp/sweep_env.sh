# bench the bf16 step under several values of one environment knob: sweep_env.sh NAME v1 v2 ...
name=$1; shift
for v in "$@"; do
  echo "== $name=$v"
  env $name=$v python bench.py --steps 20 --warmup 5 --skip-cpu --skip-roofline --skip-secondary --dtype bf16 --imsize 224 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'])"
done
