#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05b; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async_pipeline" 2>&1 | tail -15 > $O/pytest_async.txt
cat $O/pytest_async.txt
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 96"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 $B > $O/$n.json 2> $O/$n.err; echo "$n rc $?"
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'ms/round %.4f' % r['ms_per_round'], 'err', r['engine_errors'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if 'hist' not in k})
    if ap: print('   leaf hist', ap['leaf_wait_hist_us']); print('   ready hist', ap['ready_wait_hist_us'])
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1500:])
PY
}
run sync AZG_ASYNC=0
run async_default AZG_ASYNC=1
run async_128_128 AZG_ASYNC=1 AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=128
run async_160_96 AZG_ASYNC=1 AZG_ASYNC_NNET=160 AZG_ASYNC_NSEL=96
run async_128_64 AZG_ASYNC=1 AZG_ASYNC_NNET=128 AZG_ASYNC_NSEL=64
run async_64_64 AZG_ASYNC=1 AZG_ASYNC_NNET=64 AZG_ASYNC_NSEL=64
