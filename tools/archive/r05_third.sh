#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05c; mkdir -p $O
export AZG_ASYNC_TIMEOUT_MS=1500
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 96"
run() { # name, extra bench args, env...
  n=$1; x=$2; shift; shift
  env "$@" timeout 600 $B $x > $O/$n.json 2> $O/$n.err; echo "$n rc $?"
  python - <<PY
import json
try:
    r = json.load(open('$O/$n.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$n', 'value %.0f' % r['value'], 'ms/round %.4f' % r['ms_per_round'], 'err', r['engine_errors'], 'K', r['advance_every'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if 'hist' not in k})
except Exception as e:
    print('$n failed', e); print(open('$O/$n.err').read()[-1500:])
PY
}
run async_k48 "" AZG_ASYNC=1
run async_k240 "--advance-every 240" AZG_ASYNC=1
run async_k96 "--advance-every 96" AZG_ASYNC=1
cd /tmp && export TMPDIR=/tmp
AZG_ASYNC=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --roofline-rounds 96 > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/prof_summary.py /tmp/kt/kt_results.db 16 > $O/kernel_stats.md; cat $O/kernel_stats.md
N=$(python - <<PY
import sqlite3
db = sqlite3.connect('/tmp/kt/kt_results.db')
print(db.execute("select count(*) from kernels").fetchone()[0])
PY
)
python tools/prof_timeline.py /tmp/kt/kt_results.db $((N - 60)) 45 > $O/timeline.txt 2>&1; cat $O/timeline.txt
