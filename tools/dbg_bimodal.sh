# is the two-mode k_select time (49 vs 60 us) a property of the box, the process (allocation) or the clocks?  run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3 4; do
  (sleep 14; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2) &
  python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('run', $i, 'value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4))"
  wait
done
