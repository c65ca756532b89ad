cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
for w in 12 16 12 16; do echo "== AZG_V80_WAVES=$w"; AZG_V80_WAVES=$w timeout 300 python tools/time_v80.py 4096 2>&1 | grep " h2 " ; done
AZG_V80_WAVES=16 timeout 600 python -m pytest tests/test_nnet.py -x -q -m gpu -k "v80_hip or saturate or stale" 2>&1 | tail -3
