"""The oracle's Coach.executeEpisode restatement (oracle/episode.c) against episodes the REFERENCE's own Coach.executeEpisode
played (tests/golden/episode_*.npz, tools/gen_golden_episode.py): the per-ply draw order (playout-cap draw, search, move pick,
env randomness), temp_for_selfplay incl. the step schedule, random_pick's inverse CDF, which plies are recorded, the symmetry
expansion and z = roll(r, -player) -- SURVEY.md §8c G5."""
import glob
import os

import numpy as np
import pytest

import azg_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
CASES = sorted(os.path.basename(p)[len('episode_'):-len('.npz')] for p in glob.glob(os.path.join(GOLDEN, 'episode_*.npz')))


def oracle_game(variant):
    name, v = {'splendor2': (O.SPLENDOR, 2), 'santorini11': (O.SANTORINI, 11), 'azul': (O.AZUL, 0)}[variant]
    return O.OracleGame(name, v)


def episode_args(d):
    return dict(numMCTSSims=int(d['sims']), prob_fullMCTS=float(d['prob_full']), cpuct=float(d['cpuct']), fpu=float(d['fpu']),
                universes=int(d['universes']), forced_playouts=bool(d['forced']), no_mem_optim=False)


def expand_examples(og, canonical, pi, q, full, player, result):
    """Coach.py:65-69,76-82 on per-ply records: every symmetry of every full-search ply, z = roll(r, -player)"""
    out = []
    for k in np.flatnonzero(full):
        valids = og.getValidMoves(canonical[k], 0)
        for b, p, v in og.getSymmetries(canonical[k].reshape(og.shape), pi[k].astype(np.float32), valids, max_sym=128):
            out.append((b.reshape(-1), p, np.roll(result, -int(player[k])), v, q[k]))
    return out


def test_fixtures_present():
    assert len(CASES) >= 6, CASES


@pytest.mark.parametrize('case', CASES)
def test_oracle_episode_vs_reference_executeEpisode(case):
    d = np.load(os.path.join(GOLDEN, 'episode_%s.npz' % case))
    variant = case.split('_')[0]
    og = oracle_game(variant)
    o = O.run_episode(og, O.make_args(**episode_args(d)), d['init_board'], seed=int(d['seed']), stream=int(d['stream']),
                      temp=(float(d['temperature'][0]), float(d['temperature'][1])), tempThreshold=float(d['tempThreshold']))
    n = len(d['action'])
    assert o['plies'] == n
    assert np.array_equal(o['action'], d['action'])
    assert np.array_equal(o['player'], d['player'])
    assert np.array_equal(o['full'], d['full'])
    assert np.array_equal(o['canonical'], d['canonical'])
    assert np.array_equal(o['pi'], d['pi'])                 # float64 lists, bit for bit
    assert np.array_equal(o['q'], d['q'])
    ex = expand_examples(og, o['canonical'], o['pi'], o['q'], o['full'], o['player'], o['result'])
    assert len(ex) == len(d['ex_board'])
    assert np.array_equal(np.array([e[0] for e in ex]), d['ex_board'])
    assert np.array_equal(np.array([e[1] for e in ex]), d['ex_pi'].astype(np.float32))
    assert np.array_equal(np.array([e[2] for e in ex]), d['ex_z'])
    assert np.array_equal(np.array([e[3] for e in ex]).astype(np.uint8), d['ex_valid'])
    assert np.array_equal(np.array([e[4] for e in ex]), d['ex_q'])
