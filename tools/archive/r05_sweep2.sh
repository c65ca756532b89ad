#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05m; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --roofline-rounds 400"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'err', r['engine_errors'], 'budget', r['work_budget'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'descent_us', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy')})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-800:])
PY
}
export AZG_ASYNC_ADAPT=0
run b5_128 "--work-budget 5" AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run b10_128 "--work-budget 10" AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run b15_128 "--work-budget 15" AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run b10_132 "--work-budget 10" AZG_ASYNC_NNET=132 AZG_ASYNC_NSEL=124
run b10_124 "--work-budget 10" AZG_ASYNC_NNET=124 AZG_ASYNC_NSEL=132
run b10_128_again "--work-budget 10" AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
