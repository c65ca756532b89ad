"""Pins the CPU oracle (oracle/*.c) against golden vectors produced by the REFERENCE (tools/gen_golden.py) and against
the RNG-free known answers of SURVEY.md Appendix C.  CPU-only."""
import hashlib
import os

import numpy as np
import pytest

import azg_oracle as O

VARIANTS = {
    'splendor2': (O.SPLENDOR, 2), 'splendor3': (O.SPLENDOR, 3), 'splendor4': (O.SPLENDOR, 4),
    'santorini1': (O.SANTORINI, 1), 'santorini11': (O.SANTORINI, 11), 'azul': (O.AZUL, 0), 'abalone': (O.ABALONE, 0),
    'akropolis': (O.AKROPOLIS, 0), 'akropolis3': (O.AKROPOLIS, 3), 'akropolis4': (O.AKROPOLIS, 4), 'smallworld': (O.SMALLWORLD, 0), 'smallworld3': (O.SMALLWORLD, 3), 'smallworld4': (O.SMALLWORLD, 4),
}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_env_transitions(golden_dir, variant):
    """G1: valid-move masks, next states, next player, game_ended, score, round, canonical form -- bit exact."""
    d = load(golden_dir, 'env_%s.npz' % variant)
    g = O.OracleGame(*VARIANTS[variant])
    assert g.S == d['state'].shape[1] and g.A == int(d['A']) and g.P == int(d['P'])
    n = len(d['state'])
    assert n > 100
    n_seed0 = 0
    for i in range(n):
        st, pl = d['state'][i], int(d['player'][i])
        valid = g.getValidMoves(st, pl)
        assert np.array_equal(np.packbits(valid.astype(np.uint8)), d['valid'][i]), (variant, i)
        seed = int(d['seed'][i])
        rng = g.rng(injected=d['uniforms'][i]) if seed == 0 else None
        n_seed0 += seed == 0
        nb, npl = g.getNextState(st, pl, int(d['action'][i]), random_seed=seed, rng=rng)
        assert np.array_equal(nb, d['next_state'][i]), (variant, i, seed)
        assert npl == int(d['next_player'][i])
        assert np.array_equal(g.getGameEnded(nb, npl), d['ended'][i]), (variant, i)
        assert [g.getScore(nb, p) for p in range(g.P)] == list(d['score'][i])
        assert g.getRound(nb) == int(d['round'][i])
        assert np.array_equal(g.getCanonicalForm(nb, npl), d['canonical'][i])
    assert n_seed0 > 10


@pytest.mark.parametrize('variant', [v for v in VARIANTS if not v.startswith('smallworld')])      # (Smallworld's are random: test below)
def test_symmetries(golden_dir, variant):
    d = load(golden_dir, 'sym_%s.npz' % variant)
    g = O.OracleGame(*VARIANTS[variant])
    for i in range(len(d['state'])):
        syms = g.getSymmetries(d['state'][i], d['pi'][i], d['valid'][i], max_sym=128)
        assert len(syms) == int(d['count'][i])
        for k, (s, p, v) in enumerate(syms):
            assert np.array_equal(s.reshape(-1), d['out_state'][i][k]), (variant, i, k)
            assert np.array_equal(p, d['out_pi'][i][k]), (variant, i, k)
            assert np.array_equal(v.astype(np.uint8), d['out_valid'][i][k]), (variant, i, k)


def oracle_tree_digest(mc, game):
    h = hashlib.sha256()
    keys = mc.keys()
    order = sorted(range(len(keys)), key=lambda i: keys[i].tobytes())
    for i in order:
        nd = mc.node(keys[i])
        h.update(keys[i].tobytes())
        h.update(nd['Es'].tobytes())
        if nd['has_policy']:
            h.update(np.int64(nd['Ns']).tobytes())
            h.update(nd['Nsa'].tobytes())
            h.update(nd['Qsa'].tobytes())
            h.update(nd['Ps'].tobytes())
            h.update(np.float32(nd['Qs']).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


MCTS_VARIANTS = ['splendor2', 'splendor4', 'santorini1', 'santorini11', 'azul']
MCTS_SMALL = MCTS_VARIANTS + ['abalone', 'akropolis', 'smallworld']          # (no 800-simulation file for the f4 games)


@pytest.mark.parametrize('variant,typing,prefix', [(v, t, 'mcts') for v in MCTS_SMALL for t in ('numpy2', 'numba')] +
                         [(v, 'numba', 'mcts') for v in ('smallworld3', 'smallworld4', 'akropolis3', 'akropolis4')] + [(v, 'numba', 'mcts800') for v in MCTS_VARIANTS] + [('azul', 'numba', 'mcts1600')])
def test_mcts_traces(golden_dir, variant, typing, prefix):
    """G3: whole-tree parity (every node's Ns, Nsa, Qsa, Ps, Qs bit-exact through a SHA-256 digest); `mcts800` = the
    headline search size (800 simulations, tools/gen_golden_800.py); `mcts1600` = BASELINE config 5's (Azul, 1600 simulations)."""
    d = load(golden_dir, '%s_%s_%s.npz' % (prefix, variant, typing))
    g = O.OracleGame(*VARIANTS[variant])
    for i in range(len(d['case_sims'])):
        args = O.make_args(numMCTSSims=int(d['case_sims'][i]), cpuct=float(d['case_cpuct'][i]),
                           fpu=float(d['case_fpu'][i]), universes=int(d['case_universes'][i]),
                           forced_playouts=bool(d['case_forced'][i]), numpy2_scalar_typing=(typing == 'numpy2'))
        mc = O.OracleMCTS(g, args)
        probs, q, full = mc.getActionProb(d['case_root'][i], temp=1, force_full_search=True)
        nd = mc.node(d['case_root'][i])
        assert nd['Ns'] == int(d['case_Ns'][i]), (variant, i)
        assert np.array_equal(nd['Nsa'], d['case_Nsa'][i]), (variant, i)
        assert np.array_equal(nd['Qsa'], d['case_Qsa'][i]), (variant, i)
        assert nd['Qs'] == d['case_Qs'][i]
        assert np.array_equal(nd['Ps'], d['case_Ps'][i])
        assert mc.num_nodes() == int(d['case_nodes'][i])
        assert np.array_equal(probs, d['case_probs'][i])
        assert np.array_equal(q, d['case_q'][i])
        assert np.array_equal(oracle_tree_digest(mc, g), d['case_digest'][i]), (variant, i)


def test_mcts1600_azul_auto_alpha_noise(golden_dir):
    """BASELINE config 5: Azul, 1600 simulations, root Dirichlet noise with the automatic alpha = 10 / n_valid (MCTS.py:188-192) --
    the reference's own search with the sample its rng.dirichlet drew recorded; the oracle is handed the same sample."""
    d = load(golden_dir, 'mcts1600_azul_numba.npz')
    g = O.OracleGame(*VARIANTS['azul'])
    for i in range(len(d['noise_root'])):
        nv = int(d['noise_n_valid'][i])
        assert int(g.getValidMoves(d['noise_root'][i], 0).sum()) == nv and float(d['noise_alpha'][i]) == 10.0 / nv
        args = O.make_args(numMCTSSims=int(d['case_sims'][0]), cpuct=float(d['case_cpuct'][0]), fpu=float(d['case_fpu'][0]),
                           universes=int(d['case_universes'][0]), forced_playouts=bool(d['case_forced'][0]), dirichletAlpha=-1.0,
                           temperature=(1.0, 1.0, 1.0))
        mc = O.OracleMCTS(g, args, dirichlet_noise=True)
        probs, q, full = mc.getActionProb(d['noise_root'][i], temp=1, force_full_search=True, dir_noise=d['noise_sample'][i][:nv])
        nd = mc.node(d['noise_root'][i])
        assert nd['Ns'] == int(d['noise_Ns'][i]) and np.array_equal(nd['Nsa'], d['noise_Nsa'][i])
        assert np.array_equal(nd['Qsa'], d['noise_Qsa'][i]) and np.array_equal(nd['Ps'], d['noise_Ps'][i])
        assert np.array_equal(probs, d['noise_probs'][i]) and mc.num_nodes() == int(d['noise_nodes'][i])
        assert np.array_equal(oracle_tree_digest(mc, g), d['noise_digest'][i])


@pytest.mark.parametrize('variant,typing', [(v, t) for v in MCTS_SMALL for t in ('numpy2', 'numba')] + [('smallworld3', 'numba'), ('smallworld4', 'numba'), ('akropolis3', 'numba'), ('akropolis4', 'numba')])
def test_mcts_sequence_tree_reuse(golden_dir, variant, typing):
    """G3 sequence: tree reuse across moves, fast (non-full) searches, periodic clean-up (MCTS.py:86-91)."""
    d = load(golden_dir, 'mcts_%s_%s.npz' % (variant, typing))
    g = O.OracleGame(*VARIANTS[variant])
    from tools_args import MCTS_ARGS
    kw = dict(MCTS_ARGS[variant])
    args = O.make_args(numMCTSSims=int(d['seq_sims']), no_mem_optim=False, prob_fullMCTS=0.0,
                       numpy2_scalar_typing=(typing == 'numpy2'), **kw)
    mc = O.OracleMCTS(g, args)
    for i in range(len(d['seq_action'])):
        probs, q, full = mc.getActionProb(d['seq_canon'][i], temp=1, force_full_search=(i % 3 != 2), u_full=0.5)
        nd = mc.node(d['seq_canon'][i])
        assert int(full) == int(d['seq_full'][i])
        assert nd['Ns'] == int(d['seq_Ns'][i]), (variant, i)
        assert np.array_equal(nd['Nsa'], d['seq_Nsa'][i]), (variant, i)
        assert np.array_equal(probs, d['seq_probs'][i])
        assert mc.num_nodes() == int(d['seq_nodes'][i]), (variant, i)
        assert np.array_equal(oracle_tree_digest(mc, g), d['seq_digest'][i]), (variant, i)


@pytest.mark.parametrize('n', [2, 3, 4])
def test_akropolis_init_boards(golden_dir, n):
    """init_game draws the n + 2 tiles of the construction site with np.random.choice (AkropolisLogicNumba.py:294,507-508)"""
    d = load(golden_dir, 'env_akropolis%s.npz' % ('' if n == 2 else n))
    g = O.OracleGame(O.AKROPOLIS, n)
    assert g.getBoardSize() == (13, 13, 3 * n + 2) and g.A == 1014 * (n + 2) and g.P == n
    for i in range(len(d['init_boards'])):
        rng = g.rng(injected=d['init_uniforms'][i])
        assert np.array_equal(g.getInitBoard(rng).reshape(-1), d['init_boards'][i]) and rng.pos == n + 2
    assert d['score'].max() > (60 if n == 2 else 40) and set(d['seed'].tolist()) >= {0, -1, 31416}


@pytest.mark.parametrize('n', [2, 3, 4])
def test_smallworld_init_boards_and_random_symmetries(golden_dir, n):
    """init_game draws six (people, power) pairs with np.random.choice (SmallworldLogicNumba.py:1339-1356); get_symmetries shifts both
    scores by two np.random.randint offsets (:281-299), drawn here from the recorded counter streams"""
    tag = 'smallworld' if n == 2 else 'smallworld%d' % n
    d = load(golden_dir, 'env_%s.npz' % tag)
    g = O.OracleGame(O.SMALLWORLD, n)
    na = (g.A - 16) // 5
    for i in range(len(d['init_boards'])):
        rng = g.rng(injected=d['init_uniforms'][i])
        assert np.array_equal(g.getInitBoard(rng).reshape(-1), d['init_boards'][i]) and rng.pos == 12
    st = d['next_state'].reshape(-1, g.S // 8, 8)
    seen_ppl, seen_pwr = set(np.abs(st[:, na:na + 3 * n, 1]).reshape(-1).tolist()), set(np.abs(st[:, na:na + 3 * n, 2]).reshape(-1).tolist())
    assert (len(seen_ppl) == 15 and len(seen_pwr) >= 19) if n == 2 else (len(seen_ppl) >= 11 and len(seen_pwr) >= 12)
    s = load(golden_dir, 'sym_%s.npz' % tag)
    for j in range(len(s['state'])):
        rng = g.rng(seed=int(s['seed']), stream=j)
        sy = g.getSymmetries(s['state'][j], s['pi'][j], s['valids'][j], max_sym=3, rng=rng)
        assert len(sy) == int(s['count'][j]) == 3 and rng.counter == int(s['draws'][j]) == 2
        for k, (s_, p_, v_) in enumerate(sy):
            assert np.array_equal(s_.reshape(-1), s['out_state'][j, k]) and np.array_equal(p_, s['out_pi'][j, k])
            assert np.array_equal(v_, s['out_valids'][j, k].astype(bool))
