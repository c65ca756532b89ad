# AZG_SPEC_STATE sweep: 1 = state prefetched at every level, N = only below N visits of the incoming edge, 0 = never
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ab_spec
for rep in 1 2; do
for v in 1 0 2 4 8 16 64; do
  AZG_SPEC_STATE=$v python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 > gpurun_out/ab_spec/b_${v}_$rep.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_spec/b_${v}_$rep.json').read())
print('spec=$v rep=$rep value %.0f  ms/round %.4f  select_ms %.4f  frac %.4f' % (d['value'], d['ms_per_round'], d['roofline']['select_ms'], d['roofline']['frac']), flush=True)
PY
done; done
