#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async" 2>&1 | tail -4
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --roofline-rounds 400"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'whole %.0f' % r.get('value_whole_games', 0), 'plies', r['plies_completed'], 'err', r['engine_errors'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'descent_us', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy')})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-800:])
PY
}
run base "" X=1
run sync "--no-sustained" AZG_ASYNC=0
timeout 600 python bench.py --game santorini1 --steps 10 --warmup 2 --no-cpu-baseline --roofline-rounds 200 2>/dev/null | tail -1 > $O/sant.json
python -c "
import json; r=json.load(open('$O/sant.json')); print('santorini1', round(r['value']), r['engine_errors'])"
