#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -k "async" 2>&1 | tail -5
AZG_ASYNC_ADAPT=0 timeout 400 python tools/dbg_async_phases.py 2>&1 | grep -v amdgpu.ids > $O/phases_fixed.txt; tail -12 $O/phases_fixed.txt
timeout 400 python tools/dbg_async_phases.py 0 0 -1 1 2>&1 | grep -v amdgpu.ids > $O/phases_adaptive.txt; tail -27 $O/phases_adaptive.txt
