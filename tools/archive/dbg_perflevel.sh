# diagnostic: does pinning the clocks (rocm-smi --setperflevel high) remove the slow k_select mode?  (run on the GPU box; the level is reset at the end)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4))"; }
for i in 1 2 3; do run auto; done
rocm-smi --setperflevel high 2>&1 | grep -i "perf\|error\|denied" | head -3
for i in 1 2 3; do run high; done
rocm-smi --setperflevel auto 2>&1 | grep -i "perf\|error" | head -2
for i in 1 2; do run auto; done
