#!/bin/bash
# cycle breakdown of a descent call (library built with AZG_DEFINES=AZG_CYC_COUNTERS into build_ab/libazg_cyc.so): pipeline and two-kernel form
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06cyc; mkdir -p $O
for g in ${GAMES:-splendor2}; do
GAME=$g AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids > $O/cycles_pipeline_$g.txt
GAME=$g AZG_ASYNC=0 AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids > $O/cycles_two_kernel_$g.txt
echo "== $g pipeline"; cat $O/cycles_pipeline_$g.txt; echo "== $g two-kernel"; cat $O/cycles_two_kernel_$g.txt
done
