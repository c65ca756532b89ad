# what distinguishes a slow box (k_select 60 us) from a fast one (49 us)?  partition modes, clocks and power while the bench runs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i "partition"
rocm-smi --showperflevel --showmaxpower 2>/dev/null | grep -i "GPU\["
(sleep 12; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk"; rocm-smi --showpower --showtemp 2>/dev/null | grep -i "power\|junction\|memory" | head -4) &
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4))"
wait
