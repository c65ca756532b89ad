"""GPU parity: batched HIP env kernels (through the C-ABI) vs the reference's golden vectors and the pinned oracle.
Bit-exact for valid-move masks, next states, next player, game_ended, scores, round, canonical form."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = {'splendor2': ('splendor', 2), 'splendor3': ('splendor', 3), 'splendor4': ('splendor', 4),
            'santorini1': ('santorini', 1), 'santorini11': ('santorini', 11), 'azul': ('azul', 0), 'abalone': ('abalone', 0), 'akropolis': ('akropolis', 2), 'akropolis3': ('akropolis', 3), 'akropolis4': ('akropolis', 4), 'smallworld': ('smallworld', 2), 'smallworld3': ('smallworld', 3),
            'smallworld4': ('smallworld', 4)}


def make_game(variant):
    import torch
    from azg_amd import games
    name, v = VARIANTS[variant]
    assert torch.cuda.is_available()
    if name == 'azul':
        return games.AzulGame()
    if name == 'abalone':
        return games.AbaloneGame()
    if name == 'akropolis':
        return games.AkropolisGame(v)
    if name == 'smallworld':
        return games.SmallworldGame(v)
    return games.SplendorGame(v) if name == 'splendor' else games.SantoriniGame(v)


@pytest.mark.parametrize('variant', list(VARIANTS))
def test_env_vs_golden(golden_dir, variant):
    import torch
    d = np.load(os.path.join(golden_dir, 'env_%s.npz' % variant))
    g = make_game(variant)
    dev = g.device
    n = len(d['state'])
    st = torch.from_numpy(d['state']).to(dev)
    pl = torch.from_numpy(d['player'].astype(np.int32)).to(dev)
    valid = g.valid_moves_batch(st, pl).cpu().numpy()
    exp_valid = np.unpackbits(d['valid'], axis=1)[:, :g.A]
    assert np.array_equal(valid, exp_valid)
    seeds = torch.from_numpy(d['seed'].astype(np.int64)).to(dev)
    act = torch.from_numpy(d['action'].astype(np.int32)).to(dev)
    nxt_state, nxt_pl = g.next_state_batch(st, pl, act, seeds)
    det = (d['seed'] != 0) | (variant == 'abalone')           # Abalone's env step is deterministic for every seed
    assert det.sum() > 50
    assert np.array_equal(nxt_state.cpu().numpy()[det], d['next_state'][det])
    assert np.array_equal(nxt_pl.cpu().numpy(), d['next_player'].astype(np.int32))
    ns = torch.from_numpy(d['next_state']).to(dev)
    npl = torch.from_numpy(d['next_player'].astype(np.int32)).to(dev)
    ended, scores, rnd = g.game_ended_batch(ns, npl)
    assert np.array_equal(ended.cpu().numpy(), d['ended'])
    assert np.array_equal(scores.cpu().numpy(), d['score'].astype(np.int32))
    assert np.array_equal(rnd.cpu().numpy(), d['round'].astype(np.int32))
    canon = g.canonical_batch(ns, npl)
    assert np.array_equal(canon.cpu().numpy(), d['canonical'])


@pytest.mark.parametrize('variant', ['splendor2', 'splendor4', 'santorini11', 'azul', 'akropolis', 'akropolis4', 'smallworld', 'smallworld4'])
def test_true_random_moves_and_init_vs_oracle(golden_dir, variant):
    """random_seed == 0 (Coach.py:71) and Board.init_game consume the shared counter-based RNG exactly like the oracle."""
    import torch
    import azg_oracle as O
    d = np.load(os.path.join(golden_dir, 'env_%s.npz' % variant))
    g = make_game(variant)
    name, v = VARIANTS[variant]
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL, 'akropolis': O.AKROPOLIS, 'smallworld': O.SMALLWORLD}[name], v)
    dev = g.device
    n = min(len(d['state']), 400)
    st = torch.from_numpy(d['state'][:n]).to(dev)
    pl = torch.from_numpy(d['player'][:n].astype(np.int32)).to(dev)
    act = torch.from_numpy(d['action'][:n].astype(np.int32)).to(dev)
    seeds = torch.zeros(n, dtype=torch.int64, device=dev)
    counters = torch.arange(n, dtype=torch.int64, device=dev) * 3
    step = 3
    g.rng_seed = 1234
    out, nxt = g.next_state_batch(st, pl, act, seeds, stream0=77, counters=counters)
    out = out.cpu().numpy()
    cnt = counters.cpu().numpy()
    for i in range(n):
        rng = og.rng(seed=1234, stream=77 + i)
        rng.counter = 3 * i
        exp, enp = og.getNextState(d['state'][i], int(d['player'][i]), int(d['action'][i]), 0, rng)
        assert np.array_equal(out[i], exp.reshape(-1)), (variant, i)
        assert int(cnt[i]) == rng.counter
    boards = g.init_boards_batch(64, stream0=5).cpu().numpy()
    for i in range(64):
        exp = og.getInitBoard(og.rng(seed=1234, stream=5 + i)).reshape(-1)
        assert np.array_equal(boards[i], exp), (variant, i)


def test_game_py_surface():
    """The Game.py-shaped single-board methods (what main.py / pit.py / Arena call) work end to end."""
    g = make_game('splendor2')
    b = g.getInitBoard()
    assert b.shape == (56, 7) and b.dtype == np.int8
    v = g.getValidMoves(b, 0)
    assert v.dtype == bool and v.shape == (81,) and v[80]
    a = int(np.flatnonzero(v)[0])
    nb, npl = g.getNextState(b, 0, a, random_seed=31416)
    assert npl == 1 and g.getRound(nb) == 1
    assert not g.getGameEnded(nb, npl).any()
    c = g.getCanonicalForm(nb, npl)
    assert c.shape == b.shape
    assert g.stringRepresentation(b) == b.tobytes()


@pytest.mark.parametrize('variant', ['splendor2', 'splendor3', 'splendor4', 'santorini1', 'santorini11', 'azul', 'abalone', 'akropolis', 'akropolis3', 'akropolis4'])
def test_symmetries_vs_golden_and_oracle(golden_dir, variant):
    """Game.getSymmetries on device (azg_env_symmetries) vs the reference's own outputs (tests/golden/sym_*.npz) and, on
    states from random play, vs the oracle."""
    import torch
    import azg_oracle as O
    from azg_amd import games
    name, v = {'splendor2': ('splendor', 2), 'splendor3': ('splendor', 3), 'splendor4': ('splendor', 4),
               'santorini1': ('santorini', 1), 'santorini11': ('santorini', 11), 'azul': ('azul', 0), 'abalone': ('abalone', 0), 'akropolis': ('akropolis', 2), 'akropolis3': ('akropolis', 3), 'akropolis4': ('akropolis', 4), 'smallworld': ('smallworld', 2), 'smallworld3': ('smallworld', 3),
            'smallworld4': ('smallworld', 4)}[variant]
    g = {'splendor': lambda: games.SplendorGame(v), 'santorini': lambda: games.SantoriniGame(v), 'azul': games.AzulGame,
         'abalone': games.AbaloneGame, 'akropolis': lambda: games.AkropolisGame(v)}[name]()
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL, 'abalone': O.ABALONE, 'akropolis': O.AKROPOLIS}[name], v)
    K = g.max_symmetries()
    path = os.path.join(golden_dir, 'sym_%s.npz' % variant)
    if os.path.exists(path):
        d = np.load(path)
        ob, op, ov, cnt = g.symmetries_batch(torch.from_numpy(d['state'].reshape(len(d['state']), -1)).to(g.device),
                                             torch.from_numpy(d['pi']).to(g.device),
                                             torch.from_numpy(d['valid'].astype(np.uint8)).to(g.device))
        ob, op, ov, cnt = ob.cpu().numpy(), op.cpu().numpy(), ov.cpu().numpy(), cnt.cpu().numpy()
        assert np.array_equal(cnt, d['count'])
        for i in range(len(cnt)):
            k = int(cnt[i])
            assert np.array_equal(ob[i, :k], d['out_state'][i][:k].reshape(k, -1)), (variant, i)
            assert np.array_equal(op[i, :k], d['out_pi'][i][:k]), (variant, i)
            assert np.array_equal(ov[i, :k], d['out_valid'][i][:k]), (variant, i)
    # random-play states (non-empty reserves, god memos, used factories)
    rng = np.random.default_rng(3)
    states, pis, vas = [], [], []
    for gi in range(6):
        b, p = og.getInitBoard(og.rng(seed=17, stream=gi)), 0
        for ply in range(40):
            va = og.getValidMoves(b, p)
            if ply % 3 == 0:
                c = og.getCanonicalForm(b, p)
                states.append(c.reshape(-1).copy())
                vas.append(og.getValidMoves(c, 0).astype(np.uint8))
                # (Abalone's / Akropolis's get_symmetries only map the entries of valid actions: give pi the support MCTS gives it)
                pis.append(rng.random(len(va)).astype(np.float32) * (vas[-1] if name in ('abalone', 'akropolis') else 1))
            b, p = og.getNextState(b, p, int(rng.choice(np.flatnonzero(va))), random_seed=31416 + ply)
            if og.getGameEnded(b, p).any():
                break
    ob, op, ov, cnt = g.symmetries_batch(torch.from_numpy(np.stack(states)).to(g.device), torch.from_numpy(np.stack(pis)).to(g.device),
                                         torch.from_numpy(np.stack(vas)).to(g.device))
    ob, op, ov, cnt = ob.cpu().numpy(), op.cpu().numpy(), ov.cpu().numpy(), cnt.cpu().numpy()
    for i in range(len(states)):
        syms = og.getSymmetries(states[i], pis[i], vas[i], max_sym=K)
        assert len(syms) == int(cnt[i]), (variant, i)
        for k, (s, p_, v_) in enumerate(syms):
            assert np.array_equal(ob[i, k], s.reshape(-1)), (variant, i, k)
            assert np.array_equal(op[i, k], p_), (variant, i, k)
            assert np.array_equal(ov[i, k], v_.astype(np.uint8)), (variant, i, k)
    # the single-object mirror returns the reference's list-of-triples shape
    lst = g.getSymmetries(states[0].reshape(g.getBoardSize()), pis[0], vas[0].astype(bool))
    assert len(lst) == int(cnt[0]) and lst[0][0].shape == tuple(g.getBoardSize())
