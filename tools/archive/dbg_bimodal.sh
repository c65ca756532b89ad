# is the spread of the k_select time (49 ... 63 us) a property of the box, the process (allocation) or the phase of the games?  run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
FLAGS=${FLAGS:---steps 20 --warmup 5}
for i in 1 2 3 4; do
  python bench.py $FLAGS --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('run', $i, 'value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4))"
done
