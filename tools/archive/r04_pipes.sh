# round 4: XCD-pinned pipelines -- placement probe, then the driver-flag bench line for 1 / 2 / 4 / 8 / 16 groups (pinned) and 8 unpinned
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 300 python tools/dbg_placement.py > gpurun_out/r04/placement.txt 2>&1
cat gpurun_out/r04/placement.txt | tail -20
run() { # name, extra flags
  timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $2 > gpurun_out/r04/bench_$1.json 2> gpurun_out/r04/bench_$1.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04/bench_$1.json')); r = d.get('roofline') or {}
    print('$1', 'value', round(d['value']), 'ms/round', round(d['ms_per_round'], 4), 'select_ms', round(r.get('select_ms', 0), 4), 'net_ms', round((d.get('roofline_net') or {}).get('net_ms', 0), 4), 'err', d['engine_errors'], 'games', d['games_finished'])
except Exception as e:
    print('$1 FAILED', e); print(open('gpurun_out/r04/bench_$1.err').read()[-1500:])
PY
}
run g8 "--groups 8"
run g1 ""
run g4 "--groups 4"
run g16 "--groups 16"
run g2 "--groups 2"
run g8nopin "--groups 8 --no-pin-xcd"
