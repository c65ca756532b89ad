# config 3 on one box, alternating: bash tools/r06_ab_s11.sh "name[:ENV=V ...]" ...   (default: the in-tree library as it is)
cd ${GRAFT_REPO_ROOT:-/root/repo}
[ $# -eq 0 ] && set -- "base"
for i in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="X=1"
  echo -n "$name "; env $envs python bench.py --game santorini11 --steps 15 --warmup 3 --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['engine_errors'])"
done; done
