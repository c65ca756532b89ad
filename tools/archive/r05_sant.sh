#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05p; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async" 2>&1 | tail -15
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
run sync "" AZG_ASYNC=0
run async "" AZG_ASYNC=1
run async_224 "" AZG_ASYNC=1 AZG_ASYNC_NNET=224 AZG_ASYNC_NSEL=32
run async_208 "" AZG_ASYNC=1 AZG_ASYNC_NNET=208 AZG_ASYNC_NSEL=48
