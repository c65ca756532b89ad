# round 3, first GPU call: the new episode test, MFMA counters of the net, a baseline of the driver's flags on today's box
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_selfplay.py -m gpu -q -x -k "reference_executeEpisode" 2>&1 | tail -15 > $O/pytest_episode.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_VALU\b|GRBM_GUI" | head -60 > $O/counters.txt
B="python $R/bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --roofline-rounds 50"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pm -o pm -- $B > $O/pmc_mfma.log 2>&1
cd $R
python tools/prof_summary.py /tmp/pm/pm_results.db 8 > $O/pmc_mfma.md 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 1500 $O/bench_driver.json
cat $O/pytest_episode.txt
tail -30 $O/pmc_mfma.md
