cd /root/repo
for i in 1 2; do
for v in "W2=1,POL2=1" "W2=0,POL2=1:AZG_LIB=/root/repo/build_ab/libazg_w20.so" "W2=1,POL2=0:AZG_S78_POLICY2=0"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs="X=1"
  echo -n "$name "; env $envs python bench.py --game santorini11 --steps 15 --warmup 3 --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['engine_errors'])"
done; done
