# A/B of the conflict-free zero rows in the conv5 / s78 kernels + LDS conflict counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04p
python -m pytest tests/test_nnet.py -q -m gpu -k "santorini" 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -12
for r in 1 2; do for lib in "$@"; do
  echo "== $lib"; AZG_LIB=$PWD/$lib python tools/time_v89.py 2>&1 | grep "h2 us"; AZG_LIB=$PWD/$lib python tools/time_v78.py 2>&1 | grep -i "us per" | tail -2
done; done
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  AZG_LIB=$GRAFT_REPO_ROOT/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pz_$(basename $lib .so) -o pz -- python $GRAFT_REPO_ROOT/tools/time_v89.py > /dev/null 2>&1
  echo "== $lib"; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pz_$(basename $lib .so)/pz_results.db 6 | grep -E "kernel|conv5|^\|---"
done
