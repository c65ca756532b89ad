# alternating A/B/... of several libraries on one box: tools/r04_abn.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  AZG_PERCU=${AZG_PERCU:-0} AZG_LIB=$PWD/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --roofline-rounds 96 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'err', d['engine_errors'])"
done; done
