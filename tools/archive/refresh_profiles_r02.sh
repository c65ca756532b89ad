# round-2 evidence: run on the GPU box (gpurun), outputs under gpurun_out/r02/ -> copied to profiles/r02_* afterwards
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
# one ply wave of warm-up + one timed (= 1600 lock-step rounds) + 100 eager roofline rounds, opening phase
B="python $R/bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --roofline-rounds 100"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $B > $O/bench_profiled.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw -- $B > /dev/null 2>&1
cd $R
python tools/prof_summary.py /tmp/kt/kt_results.db 16 > $O/kernel_stats.md
python tools/prof_summary.py /tmp/pf/pf_results.db 8 > $O/pmc_FETCH_SIZE.md
python tools/prof_summary.py /tmp/pw/pw_results.db 8 > $O/pmc_WRITE_SIZE.md
python tools/make_traffic_json.py /tmp/pf/pf_results.db /tmp/pw/pw_results.db $O/traffic.json > /dev/null
cp $O/traffic.json profiles/r02_traffic.json
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err      # the driver's flags
python bench.py > $O/bench.json 2> $O/bench.err                                           # default: whole games
AZG_FORCE_DIST=1 AZG_BENCH_SPAWN=1 python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.log
python bench.py --prob-full 0.25 --no-cpu-baseline --no-secondary --roofline-rounds 0 2>/dev/null | tail -1 > $O/bench_mix.json
for g in azul splendor4 santorini1 santorini11; do
  python bench.py --game $g --steps $([ $g = santorini11 ] && echo 25 || echo 50) --warmup 5 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_$g.json
done
tail -c 600 $O/bench.json
cat $O/pytest.txt
