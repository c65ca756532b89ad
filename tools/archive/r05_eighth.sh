#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05h; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo rc $?
python - <<PY
import json
r = json.load(open('$O/bench_driver.json'))
print({k: r[k] for k in ('value', 'value_from_sims', 'ms_per_step', 'plies_completed', 'engine_errors', 'async_pipe', 'work_budget', 'examples_gathered')})
print(json.dumps(r.get('roofline'))[:1800])
print({k: v for k, v in (r.get('secondary') or {}).items() if k in ('value', 'error')})
print({k: v for k, v in (r.get('cpu_baseline') or {}).items() if k in ('value', 'cores')})
PY
