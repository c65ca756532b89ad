cd $GRAFT_REPO_ROOT
python bench.py --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench value', round(d['value']), 'select_ms', round(r['select_ms'],4))"
AZG_LIB=$PWD/build_ab/libazg_cyc.so GAME=splendor2 timeout 300 python tools/dbg_cycles.py 2>&1 | tail -5
