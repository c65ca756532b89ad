"""The compile-time step schedule of k_conv5_net's cell-major convolution (csrc/nn_conv5x5.hip.h C5Steps) checked on the host: the kernel
drops the (tile, tap) pairs whose tap leaves the 5 x 5 board for both cells of a tile (SantoriniNNet.py:71-84: a zero-padded 3 x 3
convolution), so the schedule must still contain every pair that touches the board, exactly once per K chunk, mark for every pair which
cell is on the board, and refill every weight fragment exactly once per kernel row."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def on_board(cell, t):
    y, x = divmod(cell, 5)
    yy, xx = y + t // 3 - 1, x + t % 3 - 1
    return cell < 25 and 0 <= yy < 5 and 0 <= xx < 5


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_conv5_step_schedule_covers_the_board(tmp_path):
    exe = str(tmp_path / 'c5_steps')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-std=c++17', '-Wno-pass-failed', '-I' + os.path.join(ROOT, 'alpha-zero-general_amd', 'csrc'),
                    '-I' + os.path.join(ROOT, 'include'), '-o', exe, os.path.join(ROOT, 'tests', 'native', 'c5_steps_dump.hip')], check=True,
                   capture_output=True, timeout=900)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split('\n')
    groups, tiles, steps, tails = {}, {}, {}, {}
    for line in out:
        w = line.split()
        if not w:
            continue
        v = [int(x) for x in w[1:]]
        if w[0] == 'G':
            groups[v[0]] = dict(nt=v[1], n=v[2], bias_at=v[3], n_tail=v[4])
        elif w[0] == 'T':
            tiles[(v[0], v[1])] = dict(rt=v[2], fin=v[3])
        elif w[0] == 'S':
            steps.setdefault(v[0], []).append(dict(s=v[1], ky=v[2], k6=v[3], ti=v[4], kind=v[5], last=v[6], epi=v[7]))
        elif w[0] == 'L':
            tails.setdefault(v[0], []).append(v[1])
    assert sorted(groups) == [0, 1, 2]
    # the 13 row tiles (two cells each; the 13th holds cell 24 and eight pad rows) are dealt out once
    assert sorted(t['rt'] for t in tiles.values()) == list(range(13))
    seen = set()
    for rg, G in groups.items():
        S = steps[rg]
        assert len(S) == G['n'] and [x['s'] for x in S] == list(range(G['n']))
        refills = {}
        for x in S:
            rt = tiles[(rg, x['ti'])]['rt']
            tap = x['ky'] * 3 + x['k6'] // 2
            a, b = on_board(2 * rt, tap), on_board(2 * rt + 1, tap)
            assert a or b, 'a kept step touches the board'
            assert x['kind'] == (0 if a and b else 1 if a else 2)
            key = (rt, tap, x['k6'] & 1)
            assert key not in seen
            seen.add(key)
            if x['last']:
                refills[(x['ky'], x['k6'])] = refills.get((x['ky'], x['k6']), 0) + 1
                assert not any(y['ky'] == x['ky'] and y['k6'] == x['k6'] for y in S[x['s'] + 1:]), 'refilled after its last use of the row'
        assert refills == {(ky, k6): 1 for ky in range(3) for k6 in range(6)}
        # every tile's epilogue is issued once, after its final step, and after the bias was requested
        issued = []
        for x in S:
            issued += [(x['s'], i) for i in range(G['nt']) if (x['epi'] >> i) & 1]
        issued += [(G['n'], i) for i in tails.get(rg, [])]
        assert sorted(i for _, i in issued) == list(range(G['nt']))
        for at, i in issued:
            fin = tiles[(rg, i)]['fin']
            assert fin == max(x['s'] for x in S if x['ti'] == i) and at > fin > G['bias_at'] - 1
    # nothing that touches the board is missing: 169 on-board (cell, tap) pairs live in 99 (tile, tap) pairs, two K chunks each
    want = {(rt, tap, c) for rt in range(13) for tap in range(9) for c in range(2) if on_board(2 * rt, tap) or on_board(2 * rt + 1, tap)}
    assert seen == want and len(want) == 198
    # the three waves of a SIMD walk the same number of steps (the convolution ends with its slowest row group)
    assert {G['n'] for G in groups.values()} == {66}
