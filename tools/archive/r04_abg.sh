# A/B of libraries for one game: tools/r04_abg.sh GAME lib1 lib2 ...
cd $GRAFT_REPO_ROOT; g=$1; shift
for r in 1 2; do for lib in "$@"; do
  AZG_LIB=$PWD/$lib timeout 900 python bench.py --game $g --steps 10 --warmup 3 --no-cpu-baseline --roofline-rounds 96 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$g $lib value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'net_ms', round(d['roofline_net']['net_ms'],4), 'err', d['engine_errors'])"
done; done
