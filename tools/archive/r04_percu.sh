cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_selfplay.py -x -q -m gpu -k "percu" 2>&1 | tail -15
run() { # name, env, extra flags
  env $2 timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $3 > gpurun_out/r04/bench_$1.json 2> gpurun_out/r04/bench_$1.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04/bench_$1.json')); r = d.get('roofline') or {}
    print('$1', 'value', round(d['value']), 'ms/round', round(d['ms_per_round'], 4), 'select_ms', round(r.get('select_ms', 0), 4), 'net_ms', round((d.get('roofline_net') or {}).get('net_ms', 0), 4), 'err', d['engine_errors'], 'games', d['games_finished'])
except Exception as e:
    print('$1 FAILED', e); print(open('gpurun_out/r04/bench_$1.err').read()[-1500:])
PY
}
run percu0 AZG_PERCU=0 ""
run percu1 AZG_PERCU=1 ""
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04/bench_percu1.json')); r = d['roofline']
print({k: r.get(k) for k in ('select_ms', 'select_wave_ms', 'net_phase_ms', 'frac', 'sims_per_launch')}, d['roofline_net']['standalone_launch_ms'])
PY
