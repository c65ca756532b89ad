cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
run() { # name, env, extra flags
  env $2 timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $3 > gpurun_out/r04/bench_$1.json 2> gpurun_out/r04/bench_$1.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04/bench_$1.json')); r = d.get('roofline') or {}
    print('$1', 'value', round(d['value']), 'ms/round', round(d['ms_per_round'], 4), 'select_ms', round(r.get('select_ms', 0), 4), 'wave', round(r.get('select_wave_ms', 0), 4), 'net', round(r.get('net_phase_ms', 0), 4), 'sims/launch', round(r.get('sims_per_launch', 0)), 'err', d['engine_errors'])
except Exception as e:
    print('$1 FAILED', e); print(open('gpurun_out/r04/bench_$1.err').read()[-800:])
PY
}
for wb in 8 10 14 20; do run p_wb$wb AZG_PERCU=1 "--work-budget $wb --roofline-rounds 96"; done
run p_k96 AZG_PERCU=1 "--advance-every 96 --roofline-rounds 96"
run p_k24 AZG_PERCU=1 "--advance-every 24 --roofline-rounds 96"
