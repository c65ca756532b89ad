# round 4: clean-up at a high-water mark -- whole games (python bench.py defaults: 70 ply waves) with the old and a smaller arena
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
run() { # name, flags
  timeout 1500 python bench.py --no-secondary --no-cpu-baseline --roofline-rounds 0 $2 > gpurun_out/r04/bench_$1.json 2> gpurun_out/r04/bench_$1.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/r04/bench_$1.json'))
    print('$1', 'value', round(d['value']), 'ms/round', round(d['ms_per_round'], 4), 'err', d['engine_errors'], 'games', d['games_finished'], 'cap', d['node_capacity'], 'max_nodes', d['max_nodes_per_tree'], 'max/cap', round(d['max_nodes_per_tree']/d['node_capacity'],3), 'live_frac', round(d['max_live_frac'],3), 'gc_runs', d['gc_runs'], 'GB', round(d['forest_bytes_per_gpu']/1e9,1))
except Exception as e:
    print('$1 FAILED', e); print(open('gpurun_out/r04/bench_$1.err').read()[-600:])
PY
}
run gc13312 ""
run gc9216 "--node-capacity 9216"
run gc8192 "--node-capacity 8192"
