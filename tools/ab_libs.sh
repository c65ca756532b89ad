# A/B of two builds on one box: build_ab/libazg_old.so (AZG_LIB) against the in-tree library, alternating, Splendor-2p and Azul
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in old new; do
  if [ $lib = old ]; then export AZG_LIB=$PWD/build_ab/libazg_old.so; else unset AZG_LIB; fi
  for g in splendor2 azul; do
  python bench.py --game $g --steps 8 --warmup 3 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib $g value', round(d['value']), 'ms/round', round(d['ms_per_round'],4), 'select_ms', round(r['select_ms'],4), 'err', d['engine_errors'])"
  done
done; done
