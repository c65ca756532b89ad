# round 4, first GPU call: wave-lifetime distribution of k_select (debug build) + the baseline bench line with the driver's flags
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
AZG_LIB=$PWD/build_ab/libazg_cyc.so timeout 600 python tools/dbg_tail2.py 1500 12 > gpurun_out/r04/tail.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err
tail -3 gpurun_out/r04/tail.txt; cut -c1-600 gpurun_out/r04/bench_base.json
