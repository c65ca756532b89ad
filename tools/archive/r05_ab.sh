# A/B two libraries on the same box: whole games (sustained leg) + the driver-flag window, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['roofline'].get('async_pipeline',{}); print('$1', 'window', round(d['value']), 'whole games', round(d.get('value_whole_games') or 0), 'errors', d.get('engine_errors'), 'descent_us', round(a.get('descent_us',0),2), 'readywait', round(a.get('ready_wait_us',0),2), 'busy', round(a.get('select_wave_busy',0),3), round(a.get('net_wg_busy',0),3))"; }
for i in 1 2; do
  for v in "$@"; do
    lib=$R/alpha-zero-general_amd/libazg_hip.so; [ $v != base ] && lib=$R/build_ab/libazg_$v.so
    AZG_LIB=$lib timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline ${AB_FLAGS:---no-sustained} 2>/dev/null | tail -1 | pr $v
  done
done
