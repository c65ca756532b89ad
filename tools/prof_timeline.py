"""Print a window of the kernel timeline of a rocprofv3 rocpd database (start offset, duration, queue/stream, name) to
see whether kernels of different streams overlap.
    python tools/prof_timeline.py results.db [first_index] [count]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    scol = 'stream_id' if 'stream_id' in cols else qcol
    rows = cur.execute("select start, end, %s, %s, name from kernels order by start" % (qcol or '0', scol or '0')).fetchall()
    i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    t0 = rows[i0][0]
    prev_end = t0
    print('columns:', cols, file=sys.stderr)
    for s, e, q, st, name in rows[i0:i0 + n]:
        print('%9.1f us  dur %7.1f  gap %+7.1f  q=%s s=%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, st,
                                                                  name[:70]))
        prev_end = max(prev_end, e)
    # overall overlap: sum of durations vs union of busy intervals
    busy = 0
    cur_s, cur_e = rows[0][0], rows[0][1]
    for s, e, *_ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('sum of kernel durations %.1f us, union of busy time %.1f us (ratio %.2f)' % (
        sum(r[1] - r[0] for r in rows) / 1e3, busy / 1e3, sum(r[1] - r[0] for r in rows) / max(busy, 1)))


def launches(path, pattern):
    """one line per dispatch whose name contains `pattern`: start offset, duration, gap since the previous matching dispatch ended"""
    db = sqlite3.connect(path)
    rows = db.execute("select start, end, name from kernels order by start").fetchall()
    rows = [r for r in rows if pattern in r[2]]
    t0 = rows[0][0]
    prev = t0
    for i, (s_, e_, name) in enumerate(rows):
        print('%4d  start %10.1f us  dur %8.1f  since previous end %+8.1f  %s' % (i, (s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - prev) / 1e3, name[:40]))
        prev = e_


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == '--launches':
        launches(sys.argv[1], sys.argv[3])
        sys.exit(0)
    main()
