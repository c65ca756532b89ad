# does the placement of the kernel-argument segment matter?  HIP_FORCE_DEV_KERNARG = 0 / 1, alternating (graph rounds and eager select time)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3 4; do for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --roofline-rounds 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v value %.0f  ms/round %.4f  select_ms %.4f  net_ms %.4f' % (d['value'], d['ms_per_round'], d['roofline']['select_ms'], d['roofline_net']['net_ms']), flush=True)"
done; done
