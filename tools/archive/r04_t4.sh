cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_env.py tests/test_gpu_mcts.py tests/test_gpu_selfplay.py -x -q -m gpu -k "santorini" 2>&1 | tail -3
bash tools/r04_abg.sh santorini1 build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so
