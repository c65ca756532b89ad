# pairs (k_select mode of the box, dependent-load latency vs footprint): is the slow mode a translation / memory-latency property?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/latency.hip -o /tmp/latency 2>/dev/null
/tmp/latency | grep -v "waves  1024\|waves 16384"
python bench.py --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4))"
