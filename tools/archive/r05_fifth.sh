#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05e; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 300 python tools/dbg_async_placement.py 2>&1 | grep -v amdgpu.ids | tee $O/placement.txt
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 96"
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
run a_128_128 "" AZG_ASYNC=1 AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run a_136_120 "" AZG_ASYNC=1 AZG_ASYNC_NNET=136 AZG_ASYNC_NSEL=120
run a_152_104 "" AZG_ASYNC=1 AZG_ASYNC_NNET=152 AZG_ASYNC_NSEL=104
run a_wait0 "" AZG_ASYNC=1 AZG_ASYNC_WAIT=0
run a_wait400 "" AZG_ASYNC=1 AZG_ASYNC_WAIT=400
