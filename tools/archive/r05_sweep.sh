#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05l; mkdir -p $O
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
run base "" X=1
run b10 "--work-budget 10" X=1
run b40 "--work-budget 40" X=1
run wait50 "" AZG_ASYNC_WAIT=50
run wait400 "" AZG_ASYNC_WAIT=400
run chunk100 "" AZG_ASYNC_CHUNK=100
run chunk800 "" AZG_ASYNC_CHUNK=800
run noadapt_128 "" AZG_ASYNC_ADAPT=0 AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run noadapt_120 "" AZG_ASYNC_ADAPT=0 AZG_ASYNC_NNET=120 AZG_ASYNC_NSEL=136
