#!/bin/bash
# the other pipeline games on the current build: bench + async counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06games; mkdir -p $O
for spec in "$@"; do
  IFS=: read -r g envs <<< "$spec"
  args=(X=1); tag=$g
  if [ -n "$envs" ]; then IFS=, read -ra ev <<< "$envs"; args+=("${ev[@]}"); tag=${g}_$(echo $envs | tr -c 'A-Za-z0-9\n' '_'); fi
  env "${args[@]}" python bench.py --game $g --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-secondary --no-sustained --roofline-rounds 200 2>$O/$tag.err | tail -1 > $O/$tag.json
  python - <<PY
import json
try:
    r = json.load(open('$O/$tag.json'))
    ap = (r.get('roofline') or {}).get('async_pipeline') or {}
    print('$tag', 'value %.0f' % r['value'], 'err', r['engine_errors'],
          {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ap.items() if k in ('n_net', 'descent_us', 'forward_us', 'leaf_wait_us', 'ready_wait_us', 'select_wave_busy', 'net_wg_busy', 'leaves_per_batch')})
except Exception as e:
    print('$tag failed', e); print(open('$O/$tag.err').read()[-600:])
PY
done
