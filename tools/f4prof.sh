# kernel-trace of the f4 bench for one game: bash tools/r04_f4prof.sh smallworld [games]
R=${GRAFT_REPO_ROOT:-/root/repo}; G=$1; N=${2:-1024}
mkdir -p $R/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kf_$G -o kt -- python $R/tools/bench_f4.py --only $G --games $N --plies 6 --net hashhip --md > $R/gpurun_out/r06p/f4prof_$G.txt 2>/dev/null
cd $R; python tools/prof_summary.py /tmp/kf_$G/kt_results.db 12 >> $R/gpurun_out/r06p/f4prof_$G.txt
cat $R/gpurun_out/r06p/f4prof_$G.txt
