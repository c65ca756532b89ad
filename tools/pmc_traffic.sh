# HBM traffic of k_select per launch (FETCH_SIZE / WRITE_SIZE, separate passes): tools/pmc_traffic.sh TAG [env assignments...]
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
O=$R/gpurun_out/traffic_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 100"
env "$@" rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$TAG -o pf -- $B > $O/bench_f.json 2>/dev/null
env "$@" rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$TAG -o pw -- $B > /dev/null 2>&1
cd $R
python tools/make_traffic_json.py /tmp/pf_$TAG/pf_results.db /tmp/pw_$TAG/pw_results.db $O/traffic.json | tail -12
python -c "
import json
d=json.load(open('$O/bench_f.json')); r=d['roofline']
print('$TAG algorithmic bytes/launch', r['bytes_per_launch'], 'select_ms(profiled)', r['select_ms'], 'traffic/algorithmic', json.load(open('$O/traffic.json'))['hbm_bytes_per_launch']/r['bytes_per_launch'])"
