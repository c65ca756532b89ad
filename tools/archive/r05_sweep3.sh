#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05n; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-secondary --roofline-rounds 0"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 900 $B $x > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    print('$n', 'value %.0f' % r['value'], 'err', r['engine_errors'], 'budget', r['work_budget'], 'plies', r['plies_completed'], 'games', r['games_finished'], 'dt %.3f' % (r['ms_per_step'] * r['steps'] / 1e3), 'sims/s %.2fM' % (r['sims_per_sec'] / 1e6))
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-800:])
PY
}
export AZG_ASYNC_ADAPT=0
run whole_b5 "--work-budget 5" X=1
run whole_b10 "--work-budget 10" X=1
run whole_b20 "--work-budget 20" X=1
run whole_b40 "--work-budget 40" X=1
run whole_sync "" AZG_ASYNC=0
