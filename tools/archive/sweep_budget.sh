cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for wb in 20 12 10 8 6; do for pre in 0 40; do
  python bench.py --steps 12 --warmup 3 --work-budget $wb --preroll-plies $pre --no-secondary --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('wb $wb pre $pre value', round(d['value']), 'from_sims', round(d['value_from_sims']), 'ms/round', round(d['ms_per_round'],4), 'err', d['engine_errors'])"
done; done; done
