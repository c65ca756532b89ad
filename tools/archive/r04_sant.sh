# round 4: Santorini / Azul device-step changes -- parity, A/B, and the rocprof evidence for the second north-star target (santorini1)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_env.py tests/test_gpu_mcts.py tests/test_gpu_selfplay.py tests/test_gpu_arena.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_sant.txt
for g in santorini1 azul santorini11; do for lib in build_ab/libazg_base.so alpha-zero-general_amd/libazg_hip.so; do
  AZG_LIB=$R/$lib timeout 900 python bench.py --game $g --steps 10 --warmup 3 --no-cpu-baseline --roofline-rounds 96 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$g $lib value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'net_ms', round(d['roofline_net']['net_ms'],4), 'err', d['engine_errors'])"
done; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --game santorini1 --steps 1 --warmup 1 --preroll-plies 0 --no-cpu-baseline --roofline-rounds 100"
rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- $B > $O/bench_profiled_santorini1.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/pfs -o pf -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pws -o pw -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pms -o pm -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pns -o pn -- $B > /dev/null 2>&1
cd $R
python tools/prof_summary.py /tmp/kts/kt_results.db 12 > $O/kernel_stats_santorini1.md
python tools/prof_summary.py /tmp/pfs/pf_results.db 6 > $O/pmc_FETCH_SIZE_santorini1.md
python tools/prof_summary.py /tmp/pws/pw_results.db 6 > $O/pmc_WRITE_SIZE_santorini1.md
python tools/prof_summary.py /tmp/pms/pm_results.db 4 | grep -E "kernel|k_conv5_net|k_select|^\|---" > $O/pmc_net_select_1_santorini1.md
python tools/prof_summary.py /tmp/pns/pn_results.db 4 | grep -E "kernel|k_conv5_net|k_select|^\|---" > $O/pmc_net_select_2_santorini1.md
python tools/make_traffic_json.py /tmp/pfs/pf_results.db /tmp/pws/pw_results.db $O/traffic_santorini1.json | tail -5
head -12 $O/kernel_stats_santorini1.md; cat $O/pmc_net_select_1_santorini1.md
