# A/B of the Santorini-with-gods net kernel: bash tools/r04_s78ab.sh lib1.so lib2.so
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nnet.py -q -m gpu -k "v78" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -6
for r in 1 2 3; do for lib in "$@"; do echo "== $lib $(AZG_LIB=$PWD/$lib python tools/time_v78.py 2>&1 | grep -i 'us per' | tail -1)"; done; done
