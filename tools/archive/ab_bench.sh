#!/bin/bash
# A/B two builds of libazg_hip.so on the SAME GPU box (boxes differ by a few %): tools/ab_bench.sh A.so B.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    AZG_LIB=$lib python bench.py --no-cpu-baseline --roofline-rounds 0 "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step']*1000,1), d['engine_errors'])"
  done
done
