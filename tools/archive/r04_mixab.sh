# A/B of a change that touches every f16 x 2 net kernel: bash tools/r04_mixab.sh lib1.so lib2.so
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nnet.py -q -m gpu 2>&1 | grep -E "^FAILED|^E  |passed|failed" | head -8
for r in 1 2; do for lib in "$@"; do
  echo "== $lib v80 $(AZG_LIB=$PWD/$lib python tools/time_v80.py 4096 2>&1 | grep ' h2 ' | head -1 | cut -c1-60) | v89 $(AZG_LIB=$PWD/$lib python tools/time_v89.py 2>&1 | grep 'h2 us') | $(AZG_LIB=$PWD/$lib python tools/time_v78.py 2>&1 | grep -i 'us per' | tail -1)"
done; done
bash tools/r04_abn.sh "$@"
for lib in "$@"; do AZG_LIB=$PWD/$lib python bench.py --game santorini1 --steps 10 --warmup 3 --no-cpu-baseline --roofline-rounds 50 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib santorini1', round(d['value']), round(d['ms_per_round'],4), d['roofline_net']['net_ms'], d['engine_errors'])"; done
