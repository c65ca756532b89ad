#!/bin/bash
# long single launches (mixed full / fast searches; Azul at 1600 simulations): error flags and the pipeline's time-out diagnostics
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06long; mkdir -p $O
run() { n=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 0 > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r=json.load(open('$O/$n.json')); print('$n', round(r['value']), 'err', r['engine_errors'], 'plies', r['plies_completed'], r.get('pipeline_timeouts'))
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1500:])
PY
}
for k in ${REPS:-1 2}; do
run mix_$k --prob-full 0.25
run azul1600_$k --game azul --sims 1600 --games 4096 --node-capacity 44000 --steps 60 --warmup 5
done
