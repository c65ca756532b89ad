"""Summarise a rocprofv3 rocpd sqlite database (kernel trace, optional PMC counters) into markdown.
    python tools/prof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/archive/r01_kernel_stats.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print('| kernel | calls | total us | avg us | min us | max us | % | vgpr | sgpr | lds B | scratch B | grid | wg |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
        print('| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s | %s |' % (
            r[0][:110].replace('|', '/'), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8], r[9], r[10],
            r[11]))
    print('\ntotal kernel time: %.1f us over %d dispatches' % (tot, sum(r[1] for r in rows)))
    try:
        pm = cur.execute("select k.name, e.counter_name, count(*), avg(e.counter_value), sum(e.counter_value) "
                         "from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                         "group by k.name, e.counter_name order by 5 desc").fetchall()
    except Exception as ex:  # schema differences
        pm = []
        print('\n(no PMC table: %s)' % ex)
    if pm:
        print('\n| kernel | counter | dispatches | avg per dispatch | sum |')
        print('|---|---|---|---|---|')
        for r in pm[:60]:
            print('| %s | %s | %d | %.3f | %.3f |' % (r[0][:90].replace('|', '/'), r[1], r[2], r[3], r[4]))


if __name__ == '__main__':
    main()
