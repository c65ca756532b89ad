#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06mix; mkdir -p $O
for spec in "$@"; do
  IFS=: read -r name lib <<< "$spec"
  args=(X=1); [ -n "$lib" ] && args+=(AZG_LIB=$R/build_ab/$lib)
  env "${args[@]}" python bench.py --prob-full 0.25 --steps ${STEPS:-70} --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 0 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    r=json.load(open('$O/$name.json')); print('$name', round(r['value']), 'err', r['engine_errors'], 'plies', r['plies_completed'])
except Exception as e:
    print('$name failed', e); print(open('$O/$name.err').read()[-1500:])
PY
done
