# A/B of tools/bench_c3.py under environment settings (one process each, same box): usage ab_c3_env.sh "A=1" "B=2 C=3" ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "== $e"
  env $e python3 tools/bench_c3.py 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print('early exits', j.pop('early_exits', None))
for k,v in j.items(): print(k, 'generate', round(v['generate_ms'],3), 'decode', round(v['decode_ms'],3), 'two_stage', round(v['two_stage_ms'],3))"
done
