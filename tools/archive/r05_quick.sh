# quick A/B: the driver-flag headline (no secondary / cpu baseline / sustained), Azul, optionally the pipeline parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); a=d['roofline'].get('async_pipeline',{}); print('$1', round(d['value']), 'errors', d.get('engine_errors'), 'descent_us', round(a.get('descent_us',0),2), 'forward_us', round(a.get('forward_us',0),2), 'busy', round(a.get('select_wave_busy',0),3), round(a.get('net_wg_busy',0),3))"; }
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-sustained 2>/dev/null | tail -1 | pr splendor2; done
timeout 600 python bench.py --game azul --steps 50 --warmup 5 --no-cpu-baseline --roofline-rounds 100 --no-sustained 2>/dev/null | tail -1 | pr azul
if [ "$1" = tests ]; then timeout 1500 python -m pytest tests/test_gpu_selfplay.py -x -q -m gpu -k "async" 2>&1 | tail -3; fi
