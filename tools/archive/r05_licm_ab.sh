#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05j; mkdir -p $O
for lib in base nnlicm; do
  [ $lib = base ] && L=$R/alpha-zero-general_amd/libazg_hip.so || L=$R/build_ab/libazg_$lib.so
  for g in santorini1 splendor4 azul santorini11; do
    AZG_LIB=$L timeout 600 python bench.py --game $g --steps 3 --warmup 1 --no-cpu-baseline --roofline-rounds 50 2>/dev/null | tail -1 > $O/${lib}_$g.json
    python - <<PY
import json
r = json.load(open('$O/${lib}_$g.json'))
print('$lib', '$g', 'value %.0f' % r['value'], 'net_ms %.4f' % (r.get('roofline_net') or {}).get('net_ms', 0), 'select_ms %.4f' % (r.get('roofline') or {}).get('select_ms', 0), 'err', r['engine_errors'])
PY
  done
  AZG_ASYNC=0 AZG_LIB=$L timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 50 2>/dev/null | tail -1 > $O/${lib}_splendor2_sync.json
  python - <<PY
import json
r = json.load(open('$O/${lib}_splendor2_sync.json'))
print('$lib', 'splendor2 sync', 'value %.0f' % r['value'], 'net_ms %.4f' % (r.get('roofline_net') or {}).get('net_ms', 0), 'select_ms %.4f' % (r.get('roofline') or {}).get('select_ms', 0), 'err', r['engine_errors'])
PY
done
