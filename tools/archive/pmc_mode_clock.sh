# is the slow k_select mode a clock (DVFS) state?  SQ_BUSY_CYCLES / kernel duration per process, several processes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 1 --warmup 1 --preroll-plies 0 --no-secondary --no-cpu-baseline --roofline-rounds 30"
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/mc$i
  rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/mc$i -o mc -- $B > /tmp/mc$i.json 2>/dev/null
  python - <<PY
import sqlite3, json
db = sqlite3.connect('/tmp/mc$i/mc_results.db'); cur = db.cursor()
for key in ('k_select', 'k_v80_net_h2'):
    dur = cur.execute("select avg(end-start)/1e3, count(*) from kernels where name like '%%%s%%'" % key).fetchone()
    out = []
    for c in ('SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'):
        r = cur.execute("select avg(e.counter_value), max(e.counter_value) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id where k.name like '%%%s%%' and e.counter_name = '%s'" % (key, c)).fetchone()
        out.append('%s avg %.0f max %.0f' % (c, r[0] or 0, r[1] or 0))
    print('run $i %-13s %.2f us (%d launches)  %s' % (key, dur[0], dur[1], '  '.join(out)), flush=True)
PY
done
