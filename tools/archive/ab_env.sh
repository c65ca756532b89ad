# A/B of one library under two environments on one box: tools/ab_env.sh "ENV_A=1" "ENV_B=0" [reps]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-4}); do for e in "$1" "$2"; do
  env $e python bench.py --steps 12 --warmup 3 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$e value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'err', d['engine_errors'])"
done; done
