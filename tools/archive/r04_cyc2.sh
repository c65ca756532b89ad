cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
AZG_PERCU=0 AZG_LIB=$PWD/build_ab/libazg_cyc.so timeout 600 python tools/dbg_cycles.py 2>&1 | tail -6
