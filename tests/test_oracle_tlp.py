"""Pins The Little Prince oracle (oracle/tlp.c) against golden vectors produced by the REFERENCE (tools/gen_golden_tlp.py):
env transitions with the uniforms the reference consumed injected in the same order, init boards, the random symmetries on a
recorded counter stream, and MCTS traces whose search-time market refills come from a recorded counter stream.  CPU-only."""
import os

import numpy as np
import pytest

import azg_oracle as O
from test_oracle_golden import oracle_tree_digest


@pytest.mark.parametrize('n', [3, 4, 5])
def test_tlp_env_transitions(golden_dir, n):
    d = np.load(os.path.join(golden_dir, 'env_tlp%d.npz' % n))
    g = O.OracleGame(O.TLP, n)
    assert g.S == d['state'].shape[1] and g.A == n * n and g.P == n and tuple(d['shape']) == g.shape
    for i in range(len(d['init_boards'])):
        rng = g.rng(injected=d['init_uniforms'][i])
        assert np.array_equal(g.getInitBoard(rng).reshape(-1), d['init_boards'][i]) and rng.pos == 1 + n
    for i in range(len(d['state'])):
        st, pl = d['state'][i], int(d['player'][i])
        assert np.array_equal(np.packbits(g.getValidMoves(st, pl).astype(np.uint8)), d['valid'][i]), i
        rng = g.rng(injected=d['uniforms'][i])
        nb, npl = g.getNextState(st, pl, int(d['action'][i]), random_seed=0, rng=rng)
        assert np.array_equal(nb.reshape(-1), d['next_state'][i]), (n, i)
        assert npl == int(d['next_player'][i]) and rng.pos == int(d['n_uniforms'][i])       # same number of draws consumed
        assert np.array_equal(g.getGameEnded(nb, npl), d['ended'][i])
        assert [g.getScore(nb, p) for p in range(n)] == list(d['score'][i]) and g.getRound(nb) == int(d['round'][i])
        assert np.array_equal(g.getCanonicalForm(nb, npl).reshape(-1), d['canonical'][i])
    assert (d['n_uniforms'] > 0).sum() >= 40 and d['ended'].any(axis=1).sum() >= 3
    assert d['score'].min() < 0                                     # the volcano penalty occurred


@pytest.mark.parametrize('n', [3, 4, 5])
def test_tlp_random_symmetries(golden_dir, n):
    """get_symmetries shuffles players / market cards / planet slots and drops duplicate states: same forms, same order, same
    number of draws as the reference on the same stream"""
    d = np.load(os.path.join(golden_dir, 'sym_tlp%d.npz' % n))
    g = O.OracleGame(O.TLP, n)
    K = d['out_state'].shape[1]
    for j in range(len(d['state'])):
        rng = g.rng(seed=int(d['seed']), stream=j)
        sy = g.getSymmetries(d['state'][j], d['pi'][j], d['valids'][j], max_sym=K, rng=rng)
        assert len(sy) == int(d['count'][j]) and rng.counter == int(d['draws'][j]), j
        for k, (s_, p_, v_) in enumerate(sy):
            assert np.array_equal(s_.reshape(-1), d['out_state'][j, k]) and np.array_equal(p_, d['out_pi'][j, k])
            assert np.array_equal(v_, d['out_valids'][j, k].astype(bool))
    assert d['count'].max() >= 4


def test_tlp_mcts_traces(golden_dir):
    d = np.load(os.path.join(golden_dir, 'mcts_tlp3_numba.npz'))
    g = O.OracleGame(O.TLP, 3)
    for i in range(len(d['case_sims'])):
        args = O.make_args(numMCTSSims=int(d['case_sims'][i]), cpuct=float(d['case_cpuct'][i]), fpu=float(d['case_fpu'][i]),
                           universes=int(d['case_universes'][i]), forced_playouts=bool(d['case_forced'][i]))
        mc = O.OracleMCTS(g, args)
        rng = g.rng(seed=int(d['case_rng_seed'][i]), stream=int(d['case_rng_stream'][i]))
        mc.set_rng(rng)
        probs, q, _ = mc.getActionProb(d['case_root'][i], temp=1, force_full_search=True)
        nd = mc.node(d['case_root'][i])
        assert rng.counter == int(d['case_rng_draws'][i]), i
        assert nd['Ns'] == int(d['case_Ns'][i]) and np.array_equal(nd['Nsa'], d['case_Nsa'][i]), i
        assert np.array_equal(nd['Qsa'], d['case_Qsa'][i]) and nd['Qs'] == d['case_Qs'][i]
        assert mc.num_nodes() == int(d['case_nodes'][i])
        assert np.array_equal(probs, d['case_probs'][i]) and np.array_equal(q, d['case_q'][i])
        assert np.array_equal(oracle_tree_digest(mc, g), d['case_digest'][i]), i
