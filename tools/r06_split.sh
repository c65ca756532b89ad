#!/bin/bash
# CU split sweep of the pipeline (n_net + n_sel = 256): bash tools/r06_split.sh 128 136 144 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
specs=()
for n in "$@"; do specs+=("net$n::AZG_ASYNC_NNET=$n,AZG_ASYNC_NSEL=$((256 - n))"); done
REPS=${REPS:-1} bash tools/r06_ab.sh "${specs[@]}"
