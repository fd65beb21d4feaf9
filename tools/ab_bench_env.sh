# A/B of bench.py's headline under environment settings (one process each): usage ab_bench_env.sh "A=1" "B=2 C=3" ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== $e"
  env $e python3 bench.py --no-cpu-baseline --no-stages --no-recall --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['value']), round(j['ms_per_step'],3), 'frac', round(r['frac'],4), 'avg_launch_ms', round(r['avg_launch_ms'],4))"
done
