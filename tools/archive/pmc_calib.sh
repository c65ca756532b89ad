# Calibrate FETCH_SIZE / WRITE_SIZE on tools/ubench/calib.hip (known byte counts), one counter per pass: -> gpurun_out/r04/pmc_calib.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib $R/tools/ubench/calib.hip || exit 1
for m in 0 1 2 3 4 5 6; do
  c=FETCH_SIZE; [ $m -eq 2 -o $m -eq 3 ] && c=WRITE_SIZE
  rocprofv3 --pmc $c -d /tmp/cal$m -o c -- /tmp/calib $m > $O/calib_mode$m.txt 2>/dev/null
  # the other counter too (what does a read pattern write, what does a write pattern fetch?)
  o=WRITE_SIZE; [ $m -eq 2 -o $m -eq 3 ] && o=FETCH_SIZE
  rocprofv3 --pmc $o -d /tmp/calo$m -o c -- /tmp/calib $m > /dev/null 2>&1
done
cd $R
python - <<'PY'
import json, sqlite3, os
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out/r04')
def per_launch(db, counter):
    cur = sqlite3.connect(db).cursor()
    row = cur.execute("select sum(e.counter_value), count(distinct e.dispatch_id) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                      "where e.counter_name = ? and k.name like '%k_calib%'", (counter,)).fetchone()
    return (row[0] or 0.0) / max(1, row[1] or 1)
out = {}
names = {0: 'dependent 1-KiB row reads', 1: '64-B reads (16 lanes x 4 B)', 2: '1-KiB row writes', 3: '16-B single-lane writes', 4: '256-B reads (64 lanes x 4 B)', 5: '128-B reads (64 lanes x 2 B)', 6: '32-B reads, all lanes the same 4 x 8 B'}
for m in range(7):
    known = json.loads(open(os.path.join(O, 'calib_mode%d.txt' % m)).read().strip().splitlines()[-1])
    main, other = ('WRITE_SIZE', 'FETCH_SIZE') if m in (2, 3) else ('FETCH_SIZE', 'WRITE_SIZE')
    a = per_launch('/tmp/cal%d/c_results.db' % m, main) * 1024.0
    b = per_launch('/tmp/calo%d/c_results.db' % m, other) * 1024.0
    out['mode%d' % m] = dict(pattern=names[m], known_bytes_per_launch=known['known_bytes_per_launch'], counter=main, counted_bytes_per_launch=a,
                            known_over_counted=known['known_bytes_per_launch'] / a if a else None, other_counter=other, other_counted_bytes_per_launch=b)
out['note'] = ('counter_value x 1024 per launch, summed over the counter instances; known_over_counted is the factor that turns the counter into '
               'bytes for that pattern (the guide prescribes x 2 for FETCH_SIZE on wide coalesced streams)')
json.dump(out, open(os.path.join(O, 'pmc_calib.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
