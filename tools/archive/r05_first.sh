#!/bin/bash
# round 5, first GPU contact of the asynchronous pipeline: the equivalence test, then short A/B bench legs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05a
export AZG_ASYNC_TIMEOUT_MS=1500
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async_pipeline" 2>&1 | tail -25 > gpurun_out/r05a/pytest_async.txt
cat gpurun_out/r05a/pytest_async.txt
for mode in 0 1; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --async-pipe $mode > gpurun_out/r05a/bench_async$mode.json 2> gpurun_out/r05a/bench_async$mode.err
  echo "mode $mode rc $?"; tail -c 3000 gpurun_out/r05a/bench_async$mode.json; tail -5 gpurun_out/r05a/bench_async$mode.err
done
