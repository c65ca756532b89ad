cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04/pytest_a.txt
bash tools/r04_abn.sh build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so
