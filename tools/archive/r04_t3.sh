cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests/test_gpu_mcts.py tests/test_gpu_selfplay.py tests/test_gpu_env.py -x -q -m gpu 2>&1 | tail -3
bash tools/r04_abn.sh build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so
