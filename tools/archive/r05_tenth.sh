#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async" 2>&1 | tail -5
timeout 400 python tools/dbg_async_phases.py 0 0 -1 1 2>&1 | grep -v amdgpu.ids > $O/phases_adaptive.txt; tail -14 $O/phases_adaptive.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_driver.json 2> $O/bench_driver.err; echo rc $?
python - <<PY
import json
r = json.load(open('$O/bench_driver.json'))
print({k: r[k] for k in ('value', 'value_from_sims', 'ms_per_step', 'plies_completed', 'engine_errors', 'async_pipe', 'work_budget', 'examples_gathered')})
PY
