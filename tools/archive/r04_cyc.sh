# round 4: where a k_select wave's cycles go today (debug build), the new net test, a symmetric CU mask probe
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
AZG_LIB=$PWD/build_ab/libazg_cyc.so timeout 600 python tools/dbg_cycles.py > gpurun_out/r04/cycles.txt 2>&1
tail -8 gpurun_out/r04/cycles.txt
timeout 600 python -m pytest tests/test_nnet.py -x -q -m gpu -k "saturate or v80_hip" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_selfplay.py tests/test_gpu_coach.py -x -q -m gpu 2>&1 | tail -5
