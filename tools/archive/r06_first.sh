#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/r06_ab.sh base:libazg_base.so glob
REPS=1 BENCH_ARGS="--games 2048" bash tools/r06_ab.sh glob2048
REPS=1 BENCH_ARGS="--games 2048" bash tools/r06_ab.sh glob2048_64:libazg_base.so:AZG_ASYNC_NNET=64,AZG_ASYNC_NSEL=64
AZG_ASYNC=0 bash tools/pmc_select.sh 2>&1 | tail -30
