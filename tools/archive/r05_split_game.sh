# split sweep for one game: GAME=splendor4 NS="40 48 56 64" bash tools/r05_split_game.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['roofline'].get('async_pipeline',{}); print('$1', round(d['value']), 'errors', d.get('engine_errors'), 'n_sel', a.get('n_sel'), 'n_net', a.get('n_net'), 'descent_us', round(a.get('descent_us',0),2), 'forward_us', round(a.get('forward_us',0),2), 'leafwait', round(a.get('leaf_wait_us',0),2), 'readywait', round(a.get('ready_wait_us',0),2), 'busy', round(a.get('select_wave_busy',0),3), round(a.get('net_wg_busy',0),3))"; }
for ns in $NS; do
  AZG_ASYNC_NSEL=$ns AZG_ASYNC_NNET=$((256-ns)) timeout 600 python bench.py --game $GAME --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --roofline-rounds 100 --no-sustained 2>/dev/null | tail -1 | pr ${GAME}_$ns
done
