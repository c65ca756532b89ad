# A/B of two libraries on one box: tools/ab_select.sh A.so B.so  (Splendor-2p driver flags, no preroll => same positions for both)
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6; do for lib in "$@"; do
  AZG_LIB=$PWD/$lib python bench.py --steps 12 --warmup 3 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'net_ms', round(d['roofline_net']['net_ms'],4), 'err', d['engine_errors'])"
done; done
