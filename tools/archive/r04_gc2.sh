cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
run() { # name, flags
  timeout 2400 python bench.py --no-secondary --no-cpu-baseline $2 > gpurun_out/r04/bench_$1.json 2> gpurun_out/r04/bench_$1.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04/bench_$1.json'))
    print('$1', 'value', round(d['value']), 'ms/round', round(d['ms_per_round'], 4), 'err', d['engine_errors'], 'games', d['games_finished'], 'cap', d['node_capacity'], 'max_nodes', d['max_nodes_per_tree'], 'max/cap', round(d['max_nodes_per_tree']/d['node_capacity'],3), 'live_frac', round(d['max_live_frac'],3), 'gc_runs', d['gc_runs'], 'GB', round(d['forest_bytes_per_gpu']/1e9,1))
except Exception as e:
    print('$1 FAILED', e); print(open('gpurun_out/r04/bench_$1.err').read()[-400:])
PY
}
run azul800 "--game azul --roofline-rounds 0"
run azul1600_c44k "--game azul --sims 1600 --games 4096 --node-capacity 44000 --steps 60 --warmup 5 --roofline-rounds 100"
run azul1600_c36k "--game azul --sims 1600 --games 4096 --node-capacity 36000 --steps 60 --warmup 5 --roofline-rounds 100"
