#!/bin/bash
# round 6: A/B of library builds on one box (pipeline, driver flags).  usage: bash tools/r06_ab.sh name[:lib.so][:ENV=V,...] ...   (in-tree library when no lib is named)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06ab; mkdir -p $O
B="python $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 400 ${BENCH_ARGS:-}"
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 $B > $O/$n.json 2> $O/$n.err
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
for rep in ${REPS:-1 2}; do
for spec in "$@"; do
  IFS=: read -r name lib envs <<< "$spec"
  args=(X=1)
  [ -n "$lib" ] && args+=(AZG_LIB=$R/build_ab/$lib)
  if [ -n "$envs" ]; then IFS=, read -ra ev <<< "$envs"; args+=("${ev[@]}"); fi
  run ${name}_$rep "${args[@]}"
done; done
