#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05f; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async_pipeline" 2>&1 | tail -15
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 480"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err; echo "$n rc $?"
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'ms/round %.4f' % r['ms_per_round'], 'err', r['engine_errors'], 'K', r['advance_every'], 'budget', r['work_budget'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if 'hist' not in k})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1500:])
PY
}
run sync "" AZG_ASYNC=0
run async "" AZG_ASYNC=1
run async_136 "" AZG_ASYNC=1 AZG_ASYNC_NNET=136 AZG_ASYNC_NSEL=120
run async_b20 "--work-budget 20" AZG_ASYNC=1
run async_b0 "--work-budget 0" AZG_ASYNC=1
