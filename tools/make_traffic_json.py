"""HBM traffic per lock-step round from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) -> profiles/<tag>_traffic.json, read by bench.py for roofline.traffic.
    python tools/make_traffic_json.py fetch.db write.db profiles/archive/r01c_traffic.json [games sims [bench_line.json [kernel]]]
bench_line.json = the JSON line the profiled command printed (same run as the FETCH_SIZE pass): its roofline block gives the
algorithmic bytes of THAT run (d, v, e, sims per launch), so that traffic / algorithmic has one denominator.
Units: the counters report KiB-ish "KB" per dispatch slice; FETCH_SIZE is doubled (gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported (uncalibrated)."""
import json
import sqlite3
import sys


def per_launch_kb(db, counter, kernels):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for k in kernels:
        row = cur.execute("select sum(e.counter_value), count(distinct e.dispatch_id) from pmc_events e join kernels k "
                          "on k.dispatch_id = e.dispatch_id where e.counter_name = ? and k.name like ?",
                          (counter, '%' + k + '<%')).fetchone()
        out[k] = (row[0] or 0.0) / max(1, row[1] or 1)
    return out


def main():
    fetch_db, write_db, dst = sys.argv[1:4]
    games = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    sims = int(sys.argv[5]) if len(sys.argv) > 5 else 800
    bench_json = sys.argv[6] if len(sys.argv) > 6 else None
    kernels = ['k_select', 'k_expand_backup']
    f = per_launch_kb(fetch_db, 'FETCH_SIZE', kernels)
    w = per_launch_kb(write_db, 'WRITE_SIZE', kernels)
    raw = (sum(f.values()) + sum(w.values())) * 1024.0
    corrected = (2.0 * sum(f.values()) + sum(w.values())) * 1024.0
    out = {'games': games, 'sims': sims, 'source': '%s + %s' % (fetch_db, write_db),
           'note': 'per lock-step round (one k_select launch with the expansion + backup in its prologue); FETCH_SIZE x 2 and WRITE_SIZE x 1: '
                   'calibrated in round 4 on tools/ubench/calib.hip (profiles/archive/r04_pmc_calib.json) -- FETCH_SIZE counts 64 B per read REQUEST, '
                   'whatever its size: a wave instruction that touches >= 128 contiguous bytes (a record\'s hot run, a child-slot run, the '
                   'action ids, a state chunk) is counted at half its bytes, a 64-B request exactly, a 32-B record header at twice its '
                   'bytes (64 B are fetched); WRITE_SIZE is exact for full lines and counts 32 B for a 16-B store.  x 2 on all fetches is '
                   'therefore an UPPER bound for k_select (exact for its wide runs, which are most of its requests)',
           'fetch_kb': f, 'write_kb': w, 'hbm_bytes_per_launch_raw': raw, 'hbm_bytes_per_launch': corrected}
    if bench_json:
        try:
            lines = [l for l in open(bench_json).read().splitlines() if l.startswith('{')]
            r = json.loads(lines[-1])['roofline']
            out['profiled_run'] = {k: r.get(k) for k in ('d_levels_per_sim', 'v_valid_per_level', 'e_expansions_per_sim', 'bytes_per_sim',
                                                          'sims_per_launch', 'bytes_per_launch', 'select_ms')}
            out['traffic_over_algorithmic'] = corrected / r['bytes_per_launch']
            out['traffic_over_algorithmic_lower_bound'] = raw / r['bytes_per_launch']
        except Exception as ex:
            out['profiled_run'] = 'unreadable: %r' % (ex,)
    json.dump(out, open(dst, 'w'), indent=1)
    print(open(dst).read())


if __name__ == '__main__':
    main()
