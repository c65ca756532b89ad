#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05q; mkdir -p $O
B="python $R/bench.py --game santorini1 --steps 8 --warmup 2 --no-cpu-baseline --roofline-rounds 200"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'err', r['engine_errors'], 'budget', r['work_budget'], 'plies', r['plies_completed'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'n_sel', 'descent_us', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy', 'leaves_per_batch')})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1200:])
PY
}
run a_204 "" AZG_ASYNC_NNET=204 AZG_ASYNC_NSEL=52
run a_200 "" AZG_ASYNC_NNET=200 AZG_ASYNC_NSEL=56
run a_196 "" AZG_ASYNC_NNET=196 AZG_ASYNC_NSEL=60
run a_204_b10 "--work-budget 10" AZG_ASYNC_NNET=204 AZG_ASYNC_NSEL=52
run a_204_b40 "--work-budget 40" AZG_ASYNC_NNET=204 AZG_ASYNC_NSEL=52
run a_204_wait0 "" AZG_ASYNC_NNET=204 AZG_ASYNC_NSEL=52 AZG_ASYNC_WAIT=0
