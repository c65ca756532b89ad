cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nnet.py -x -q -m gpu 2>&1 | tail -3
for lib in build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so; do AZG_LIB=$PWD/$lib timeout 300 python tools/time_v89.py 2>&1 | tail -4; done
bash tools/r04_abg.sh santorini1 build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so
