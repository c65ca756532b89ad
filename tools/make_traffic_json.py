"""HBM traffic per lock-step round from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) -> profiles/<tag>_traffic.json, read by bench.py for roofline.traffic.
    python tools/make_traffic_json.py fetch.db write.db profiles/r01c_traffic.json [games sims]
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
    kernels = ['k_select', 'k_expand_backup']
    f = per_launch_kb(fetch_db, 'FETCH_SIZE', kernels)
    w = per_launch_kb(write_db, 'WRITE_SIZE', kernels)
    raw = (sum(f.values()) + sum(w.values())) * 1024.0
    corrected = (2.0 * sum(f.values()) + sum(w.values())) * 1024.0
    json.dump({'games': games, 'sims': sims, 'source': '%s + %s' % (fetch_db, write_db),
               'note': 'per lock-step round (one k_select + one k_expand_backup launch); FETCH_SIZE doubled per the gfx950 '
                       'correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated, taken as reported',
               'fetch_kb': f, 'write_kb': w, 'hbm_bytes_per_launch_raw': raw, 'hbm_bytes_per_launch': corrected},
              open(dst, 'w'), indent=1)
    print(open(dst).read())


if __name__ == '__main__':
    main()
