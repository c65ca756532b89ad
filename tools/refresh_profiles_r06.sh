# round-6 evidence: run on the GPU box (gpurun), outputs under gpurun_out/r06prof/ -> copied to profiles/r06_* afterwards
# usage: bash tools/refresh_profiles_r06.sh [part ...]   parts: tests bench prof sant sant11 games variants phases v78 (default: all)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06prof; mkdir -p $O
PARTS=${@:-tests bench prof sant sant11 games variants phases v78}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
cd $R
if has tests; then python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_tail.txt; tail -3 $O/pytest_tail.txt > $O/pytest.txt; fi
if has bench; then
  python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err      # the driver's flags (pipeline; whole-games leg; Santorini; CPU baseline)
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 > $O/bench_driver_repeat.json
  AZG_ASYNC=0 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_two_kernel_rounds.json   # the round-4 form, same box
  python bench.py --steps 20 --warmup 5 --preroll-plies 0 --no-secondary --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 > $O/bench_driver_opening.json
  python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err                         # default: whole games
  AZG_FORCE_DIST=1 AZG_BENCH_SPAWN=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-sustained > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.log
  python bench.py --prob-full 0.25 --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 0 2>/dev/null | tail -1 > $O/bench_mix.json
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  # the pipeline under the kernel trace: the two persistent kernels side by side (driver flags, shorter window)
  B="python $R/bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --no-sustained --roofline-rounds 400"
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $B > $O/bench_profiled.json 2>/dev/null
  cd $R
  python tools/prof_summary.py /tmp/kt/kt_results.db 12 > $O/kernel_stats.md
  python tools/prof_timeline.py /tmp/kt/kt_results.db --launches k_async > $O/timeline_pipeline.txt 2>&1
  # counters need serialised kernels, which the pipeline's two concurrent kernels cannot have: the traffic / instruction counters are
  # those of the SAME descent and forward code in their two-kernel form (AZG_ASYNC=0), opening phase as in rounds 2-4
  cd /tmp
  B0="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --no-sustained --roofline-rounds 100"
  AZG_ASYNC=0 rocprofv3 --kernel-trace --stats -d /tmp/kt0 -o kt -- $B0 > $O/bench_profiled_two_kernel.json 2>/dev/null
  AZG_ASYNC=0 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf -- $B0 > $O/bench_profiled_fetch.json 2>/dev/null
  AZG_ASYNC=0 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw -- $B0 > /dev/null 2>&1
  # (at most four counters per pass: eight in one pass crashed rocprofv3 on this round's boxes)
  AZG_ASYNC=0 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA -d /tmp/pm -o pm -- $B0 > /dev/null 2>&1
  AZG_ASYNC=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/pm2 -o pm -- $B0 > /dev/null 2>&1
  AZG_ASYNC=0 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/pn -o pn -- $B0 > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py /tmp/kt0/kt_results.db 12 > $O/kernel_stats_two_kernel.md
  python tools/prof_summary.py /tmp/pf/pf_results.db 8 > $O/pmc_FETCH_SIZE.md
  python tools/prof_summary.py /tmp/pw/pw_results.db 8 > $O/pmc_WRITE_SIZE.md
  python tools/prof_summary.py /tmp/pm/pm_results.db 4 | grep -E "^\| kernel|k_v80_net_h2|k_select|^\|---" > $O/pmc_net_select_1.md
  python tools/prof_summary.py /tmp/pm2/pm_results.db 4 | grep -E "^\| kernel|k_v80_net_h2|k_select|^\|---" > $O/pmc_net_select_3.md
  python tools/prof_summary.py /tmp/pn/pn_results.db 4 | grep -E "^\| kernel|k_v80_net_h2|k_select|^\|---" > $O/pmc_net_select_2.md
  python tools/make_traffic_json.py /tmp/pf/pf_results.db /tmp/pw/pw_results.db $O/traffic.json 4096 800 $O/bench_profiled_fetch.json > /dev/null
fi
if has sant; then
  python bench.py --game santorini1 --steps 10 --warmup 2 --no-cpu-baseline --roofline-rounds 200 2>/dev/null | tail -1 > $O/bench_santorini1.json
  AZG_ASYNC=0 python bench.py --game santorini1 --steps 10 --warmup 2 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_santorini1_two_kernel_rounds.json
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python $R/bench.py --game santorini1 --steps 2 --warmup 1 --no-cpu-baseline --roofline-rounds 200 > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py /tmp/kts/kt_results.db 10 > $O/kernel_stats_santorini1.md
fi
if has sant11; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d /tmp/kts11 -o kt -- python $R/bench.py --game santorini11 --steps 3 --warmup 1 --no-cpu-baseline --roofline-rounds 100 > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py /tmp/kts11/kt_results.db 10 > $O/kernel_stats_santorini11.md
fi
if has games; then
  for g in azul splendor4 santorini11; do
    python bench.py --game $g --steps $([ $g = santorini11 ] && echo 25 || echo 50) --warmup 5 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_$g.json
  done
  python bench.py --game azul --sims 1600 --games 4096 --node-capacity 44000 --steps 60 --warmup 5 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_azul1600.json
fi
if has variants; then
  AZG_PERCU=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 > $O/bench_percu.json
  AZG_ASYNC_SHARED=0 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 > $O/bench_per_tree_budget.json
fi
if has phases; then
  python tools/dbg_async_phases.py 2>&1 | grep -v amdgpu.ids > $O/phases.txt
  AZG_ASYNC=1 python tools/dbg_async_placement.py 2>&1 | grep -v amdgpu.ids > $O/placement_pipeline.txt
  [ -f build_ab/libazg_cyc.so ] && AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids > $O/cycles_pipeline.txt
  [ -f build_ab/libazg_cyc.so ] && AZG_ASYNC=0 AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_cycles.py 2>&1 | grep -v amdgpu.ids > $O/cycles_two_kernel.txt
fi
if has v78; then
  python tools/time_v78.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/v78_forward.txt
  [ -f build_ab/libazg_ph.so ] && AZG_LIB=$R/build_ab/libazg_ph.so python tools/dbg_nn_phases_s78.py 2>&1 | grep -v amdgpu.ids >> $O/v78_forward.txt
fi
ls -la $O | tail -50
