"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd sqlite): for every ordered pair (previous kernel ->
next kernel) the distribution of start(next) - end(previous).  python tools/prof_gaps.py /tmp/kt/kt_results.db"""
import sqlite3
import sys
from collections import defaultdict


def short(n):
    for key in ('k_select', 'k_v80_net_h2', 'k_selfplay_advance', 'k_root_noise', 'k_gc', 'k_after_gc', 'k_conv5_net', 'k_mb1d_net'):
        if key in n:
            return key
    return n[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
    gaps = defaultdict(list)
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
    print('| previous -> next | pairs | median gap us | mean us | p90 us |')
    print('|---|---|---|---|---|')
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:10]:
        v = sorted(v)
        print('| %s -> %s | %d | %.2f | %.2f | %.2f |' % (k[0], k[1], len(v), v[len(v) // 2], sum(v) / len(v), v[int(len(v) * 0.9)]))
    sel = [(s, e) for n, s, e in rows if 'k_select' in n]
    if len(sel) > 200:
        mid = sel[100:-100]
        per = (mid[-1][0] - mid[0][0]) / 1e3 / (len(mid) - 1)
        print('\nk_select start-to-start over the middle %d launches: %.2f us per round (kernel time of select %.2f us)' % (
            len(mid), per, sum(e - s for s, e in mid) / 1e3 / len(mid)))


if __name__ == '__main__':
    main()
