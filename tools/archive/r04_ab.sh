# A/B of two libraries on one box (alternating), Splendor-2p driver flags: tools/r04_ab.sh A.so B.so [extra bench flags]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for r in 1 2 3; do for lib in $A $B; do
  AZG_PERCU=${AZG_PERCU:-0} AZG_LIB=$PWD/$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --roofline-rounds 96 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'net_ms', round(d['roofline_net']['net_ms'],4), 'err', d['engine_errors'])"
done; done
