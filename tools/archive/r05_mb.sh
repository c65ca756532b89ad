#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05t; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async" 2>&1 | tail -15
run() { # name, game, extra bench args, env...
  n=$1; g=$2; x=$3; shift; shift; shift
  env "$@" timeout 600 python $R/bench.py --game $g --steps 10 --warmup 2 --no-cpu-baseline --roofline-rounds 200 $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'err', r['engine_errors'], 'async', r['async_pipe'], 'plies', r['plies_completed'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'n_sel', 'descent_us', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy', 'leaves_per_batch')})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1200:])
PY
}
run azul_sync azul "" AZG_ASYNC=0
run azul_async azul "" AZG_ASYNC=1
run azul_async_64 azul "" AZG_ASYNC=1 AZG_ASYNC_NNET=64 AZG_ASYNC_NSEL=192
run azul_async_128 azul "" AZG_ASYNC=1 AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run spl4_sync splendor4 "" AZG_ASYNC=0
run spl4_async splendor4 "" AZG_ASYNC=1
run spl4_async_192 splendor4 "" AZG_ASYNC=1 AZG_ASYNC_NNET=192 AZG_ASYNC_NSEL=64
