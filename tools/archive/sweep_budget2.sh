# work budget x advance cadence, driver-like flags (timings repeat to 0.1 % since the dispatch-packet fix)
cd $GRAFT_REPO_ROOT
for wb in ${WBS:-10 12 14 16}; do for ae in ${AES:-32 48 64}; do
  python bench.py --steps 12 --warmup 3 --work-budget $wb --advance-every $ae --no-secondary --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('wb $wb adv $ae value', round(d['value']), 'from_sims', round(d['value_from_sims']), 'ms/round', round(d['ms_per_round'],4), 'err', d['engine_errors'])"
done; done
