#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -x -q 2>&1 | tail -15
echo "== cycles async"; AZG_LIB=$R/build_ab/libazg_cyc.so timeout 300 python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids | tee $O/cycles_async.txt
echo "== cycles sync";  AZG_ASYNC=0 AZG_LIB=$R/build_ab/libazg_cyc.so timeout 300 python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids | tee $O/cycles_sync.txt
echo "== pmc on the pipeline"
cd /tmp && export TMPDIR=/tmp
AZG_ASYNC_TIMEOUT_MS=200 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-sustained --no-cpu-baseline --roofline-rounds 100 > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo rc $?
cd $R; python tools/prof_summary.py /tmp/pf/pf_results.db 6 2>&1 | tail -12 | cut -c1-200
python - <<PY
import json
try:
    r = json.load(open('$O/pmc_fetch.json')); print('pmc run: value %.0f errors %d' % (r['value'], r['engine_errors']))
except Exception as e:
    print('pmc run produced no line', e); print(open('$O/pmc_fetch.err').read()[-600:])
PY
