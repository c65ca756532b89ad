cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mcts.py -x -q -m gpu 2>&1 | tail -2
