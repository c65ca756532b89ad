cd $GRAFT_REPO_ROOT
for wb in 10 20; do for k in 16 48; do
  timeout 900 python bench.py --game santorini1 --steps 10 --warmup 3 --no-cpu-baseline --roofline-rounds 0 --work-budget $wb --advance-every $k 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('santorini1 wb $wb K $k value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'err', d['engine_errors'])"
done; done
