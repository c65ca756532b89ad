# round-4 evidence: run on the GPU box (gpurun), outputs under gpurun_out/r04p/ -> copied to profiles/r04_* afterwards
# usage: bash tools/refresh_profiles_r04.sh [part ...]   parts: tests bench prof sant games variants nets tail f4 (default: all)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p; mkdir -p $O
PARTS=${@:-tests bench prof sant games variants nets tail f4}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
cd $R
if has tests; then python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest.txt; fi
if has bench; then
  python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err      # the driver's flags
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_repeat.json
  python bench.py --steps 20 --warmup 5 --preroll-plies 0 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_opening.json
  python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err                         # default: whole games
  AZG_FORCE_DIST=1 AZG_BENCH_SPAWN=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.log
  python bench.py --prob-full 0.25 --no-cpu-baseline --no-secondary --roofline-rounds 0 2>/dev/null | tail -1 > $O/bench_mix.json
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  # one ply wave of warm-up + one timed (= 1600 lock-step rounds) + 100 eager roofline rounds, opening phase (no pre-roll: same positions as r02 / r03)
  B="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 100"
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $B > $O/bench_profiled.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf -- $B > $O/bench_profiled_fetch.json 2>/dev/null
  rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o pm -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pn -o pn -- $B > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py /tmp/kt/kt_results.db 16 > $O/kernel_stats.md
  python tools/prof_summary.py /tmp/pf/pf_results.db 8 > $O/pmc_FETCH_SIZE.md
  python tools/prof_summary.py /tmp/pw/pw_results.db 8 > $O/pmc_WRITE_SIZE.md
  python tools/prof_summary.py /tmp/pm/pm_results.db 4 | grep -E "^\| kernel|k_v80_net_h2|k_select|^\|---" > $O/pmc_net_select_1.md
  python tools/prof_summary.py /tmp/pn/pn_results.db 4 | grep -E "^\| kernel|k_v80_net_h2|k_select|^\|---" > $O/pmc_net_select_2.md
  python tools/make_traffic_json.py /tmp/pf/pf_results.db /tmp/pw/pw_results.db $O/traffic.json 4096 800 $O/bench_profiled_fetch.json > /dev/null
  python tools/prof_gaps.py /tmp/kt/kt_results.db > $O/kernel_gaps.md 2>/dev/null
fi
if has games; then
  for g in azul splendor4 santorini1 santorini11; do
    python bench.py --game $g --steps $([ $g = santorini11 ] && echo 25 || echo 50) --warmup 5 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_$g.json
  done
  python bench.py --game azul --sims 1600 --games 4096 --node-capacity 44000 --steps 60 --warmup 5 --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > $O/bench_azul1600.json
fi
if has variants; then
  # the measured-and-dropped round structures: the per-CU round kernel, two / four independent pipelines
  AZG_PERCU=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_percu.json
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --groups 2 2>/dev/null | tail -1 > $O/bench_groups2.json
  python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --groups 4 2>/dev/null | tail -1 > $O/bench_groups4.json
  cd /tmp && export TMPDIR=/tmp
  AZG_PERCU=1 rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt -- python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 96 > /dev/null 2>&1
  cd $R; python tools/prof_summary.py /tmp/ktp/kt_results.db 8 > $O/kernel_stats_percu.md
fi
if has nets; then
  for w in 12 16; do echo "== AZG_V80_WAVES=$w"; AZG_V80_WAVES=$w python tools/time_v80.py 4096 2>&1 | grep " h2 "; done > $O/time_v80.txt 2>&1
  python tools/time_v89.py > $O/time_v89.txt 2>&1
  python tools/time_v78.py > $O/time_v78.txt 2>&1
fi
if has tail; then
  [ -f build_ab/libazg_cyc.so ] && AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_tail2.py 1500 12 > $O/tail.txt 2>&1
  [ -f build_ab/libazg_cyc.so ] && AZG_LIB=$R/build_ab/libazg_cyc.so python tools/dbg_cycles.py > $O/cycles.txt 2>&1
  python tools/dbg_placement.py > $O/placement.txt 2>&1
fi
if has sant; then
  # the second north-star target: kernel trace + counters of Santorini no-gods (opening plies)
  cd /tmp && export TMPDIR=/tmp
  B="python $R/bench.py --game santorini1 --steps 1 --warmup 1 --preroll-plies 0 --no-cpu-baseline --roofline-rounds 100"
  rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- $B > $O/bench_profiled_santorini1.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pfs -o pf -- $B > $O/bench_profiled_fetch_santorini1.json 2>/dev/null
  rocprofv3 --pmc WRITE_SIZE -d /tmp/pws -o pw -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pms -o pm -- $B > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pns -o pn -- $B > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py /tmp/kts/kt_results.db 12 > $O/kernel_stats_santorini1.md
  python tools/prof_summary.py /tmp/pfs/pf_results.db 6 > $O/pmc_FETCH_SIZE_santorini1.md
  python tools/prof_summary.py /tmp/pws/pw_results.db 6 > $O/pmc_WRITE_SIZE_santorini1.md
  python tools/prof_summary.py /tmp/pms/pm_results.db 4 | grep -E "^\| kernel|k_conv5_net|k_select|^\|---" > $O/pmc_net_select_1_santorini1.md
  python tools/prof_summary.py /tmp/pns/pn_results.db 4 | grep -E "^\| kernel|k_conv5_net|k_select|^\|---" > $O/pmc_net_select_2_santorini1.md
  python tools/make_traffic_json.py /tmp/pfs/pf_results.db /tmp/pws/pw_results.db $O/traffic_santorini1.json 4096 800 $O/bench_profiled_fetch_santorini1.json > /dev/null
fi
if has f4; then
  bash tools/r04_f4ab.sh > /dev/null 2>&1          # -> $O/f4_bench.md: hash (torch ops) / hashhip (one engine kernel) / mlp (TorchModuleEvaluator)
  for g in smallworld botanik; do python tools/bench_f4.py --md --plies 12 --net hashhip --games 4096 --only $g 2>/dev/null; done >> $O/f4_bench.md
fi
ls -la $O | tail -40
