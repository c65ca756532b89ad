cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-secondary --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'err', d['engine_errors'], flush=True)"; }
for v in 8 1 0 3 16 32; do echo -n "AZG_SPEC_STATE=$v: "; AZG_SPEC_STATE=$v run; done
for cfg in "11 48" "9 48" "10 40" "10 56"; do set -- $cfg; echo -n "wb $1 adv $2: "; run --work-budget $1 --advance-every $2; done
