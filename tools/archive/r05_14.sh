#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05v; mkdir -p $O
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 400"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'plies', r['plies_completed'], 'err', r['engine_errors'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'descent_us', 'descent_cycles', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy')})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-800:])
PY
}
run w16 "" X=1
run w12_128 "" AZG_LIB=$R/build_ab/libazg_w12.so
run w12_112 "" AZG_LIB=$R/build_ab/libazg_w12.so AZG_ASYNC_NNET=112 AZG_ASYNC_NSEL=144
run w12_96 "" AZG_LIB=$R/build_ab/libazg_w12.so AZG_ASYNC_NNET=96 AZG_ASYNC_NSEL=160
