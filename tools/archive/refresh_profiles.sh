set -x
cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r01c_pytest.txt
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 400 --warmup 100 --no-cpu-baseline --roofline-rounds 100"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $B > /root/repo/gpurun_out/r01c_bench_profiled.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw -- $B > /dev/null 2>&1
cd /root/repo
python tools/prof_summary.py /tmp/kt/kt_results.db 16 > gpurun_out/r01c_kernel_stats.md
python tools/prof_summary.py /tmp/pf/pf_results.db 8 > gpurun_out/r01c_pmc_FETCH_SIZE.md
python tools/prof_summary.py /tmp/pw/pw_results.db 8 > gpurun_out/r01c_pmc_WRITE_SIZE.md
python tools/make_traffic_json.py /tmp/pf/pf_results.db /tmp/pw/pw_results.db gpurun_out/r01c_traffic.json > /dev/null
cp gpurun_out/r01c_traffic.json profiles/r01c_traffic.json
python bench.py > gpurun_out/r01c_bench.json 2> gpurun_out/r01c_bench.err
python bench.py --steps 1700 --warmup 100 --no-cpu-baseline > gpurun_out/r01c_bench_fresh.json 2>/dev/null
python bench.py --prob-full 0.25 --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 > gpurun_out/r01c_bench_mix.json
python bench.py --steps 3000 --warmup 500 --roofline-rounds 0 --cpu-procs 64 2>/dev/null | tail -1 > gpurun_out/r01c_bench_cpu64.json
for g in azul splendor4 santorini1 santorini11; do
  python bench.py --game $g --steps $([ $g = santorini11 ] && echo 20000 || echo 40000) --warmup 4000 --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 > gpurun_out/r01c_bench_$g.json
done
tail -c 1500 gpurun_out/r01c_bench.json
cat gpurun_out/r01c_pytest.txt
