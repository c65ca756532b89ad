# A/B of V80 net kernel builds: bash tools/r04_v80ab.sh lib1.so lib2.so ...   (parity tests run on the tree's own library)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nnet.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | head -5
python -m pytest tests/test_gpu_selfplay.py -q -m gpu -k "percu" 2>&1 | grep -E "^FAILED|passed|failed" | head -5
for r in 1 2 3; do for lib in "$@"; do echo "== $lib $(AZG_LIB=$PWD/$lib python tools/time_v80.py 4096 2>&1 | grep ' h2 ' | head -1)"; done; done
bash tools/r04_abn.sh "$@"
