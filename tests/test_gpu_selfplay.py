"""GPU parity: device-resident self-play (Coach.executeEpisode on the forest) vs the oracle's episode with the same
counter-based RNG stream and the deterministic hash-net: same moves, same examples (board, pi, z, q)."""
import numpy as np
import pytest

from tools_args import MCTS_ARGS

pytestmark = pytest.mark.gpu


class Args(dict):
    __getattr__ = dict.get


def _play_until(f, T, games, mode, chunk=256, max_iter=200000):
    """drive forest `f` with the hash-net until `games` games have ended.  mode 'rounds': the host-driven lock-step rounds (select ->
    predict -> expand_backup -> selfplay_advance); mode 'async': the asynchronous tree pipeline ITSELF (persistent descent kernel + the
    hash-net inside the pipeline's persistent evaluator kernel, azg_forest_async_rounds_hashnet; exactly `chunk` calls per tree per launch)"""
    import torch
    from hashnet import HashNetPipeline, HashNetTorch
    shape = f.board_shape()
    if mode == 'async':
        net = HashNetPipeline(f.P)
        pi = torch.zeros((T, f.A), dtype=torch.float32, device='cuda')
        v = torch.zeros((T, f.P), dtype=torch.float32, device='cuda')
        for _ in range(max_iter // chunk + 1):
            f.async_rounds_v80(net, pi, v, chunk, shared_budget=False)
            st = f.stats()
            assert st['errors'] == 0, st
            if st['games'] >= games:
                break
        return st
    net = HashNetTorch(f.P)
    for rnd in range(max_iter):
        f.select()
        pi, vv = net.predict_batch(f.leaf_states.view((T,) + shape), f.leaf_valid.bool())
        f.expand_backup(pi, vv)
        f.selfplay_advance()
        if rnd % chunk == chunk - 1:
            st = f.stats()
            assert st['errors'] == 0, st
            if st['games'] >= games:
                break
    return f.stats()



_ASYNC_GAMES = ('splendor2', 'splendor4', 'santorini1', 'azul')     # games with a descent kernel in the pipeline


@pytest.mark.parametrize('variant,prob_full', [('splendor2', 1.0), ('splendor2', 0.5), ('splendor4', 1.0), ('santorini1', 1.0),
                                                ('santorini11', 1.0), ('azul', 1.0), ('abalone', 1.0), ('akropolis', 1.0), ('akropolis3', 1.0), ('akropolis4', 1.0), ('smallworld', 1.0), ('smallworld3', 1.0), ('smallworld4', 1.0)])
def test_selfplay_first_games_vs_oracle(variant, prob_full):
    _first_games_vs_oracle(variant, prob_full, 'rounds')


@pytest.mark.parametrize('variant,prob_full', [('splendor2', 1.0), ('splendor2', 0.5), ('splendor4', 1.0), ('santorini1', 1.0), ('azul', 1.0)])
def test_async_pipeline_first_games_vs_oracle(variant, prob_full):
    """the PIPELINE itself (not its two-kernel twin) against the oracle's episodes: persistent descent kernel, the hash-net evaluated inside
    the pipeline's persistent evaluator kernel, moves / examples / restarts / clean-ups in-kernel (Coach.py:37-84,117-144)"""
    _first_games_vs_oracle(variant, prob_full, 'async')


def _first_games_vs_oracle(variant, prob_full, mode):
    import torch
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    name, v = {'splendor2': ('splendor', 2), 'splendor4': ('splendor', 4), 'santorini1': ('santorini', 1),
               'santorini11': ('santorini', 11), 'azul': ('azul', 0), 'abalone': ('abalone', 0), 'akropolis': ('akropolis', 2), 'akropolis3': ('akropolis', 3), 'akropolis4': ('akropolis', 4), 'smallworld': ('smallworld', 2), 'smallworld3': ('smallworld', 3),
               'smallworld4': ('smallworld', 4)}[variant]
    g = {'splendor': lambda: games.SplendorGame(v), 'santorini': lambda: games.SantoriniGame(v), 'azul': games.AzulGame,
         'abalone': games.AbaloneGame, 'akropolis': lambda: games.AkropolisGame(v), 'smallworld': lambda: games.SmallworldGame(v)}[name]()
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL, 'abalone': O.ABALONE, 'akropolis': O.AKROPOLIS, 'smallworld': O.SMALLWORLD}[name], v)
    from conftest import poison_onchip
    poison_onchip(0x7FC00000)
    kw = dict(MCTS_ARGS[variant])
    sims, T, seed, stream0 = 40, 16, 4242, 1000
    if name == 'akropolis':
        # hundreds of valid placements: with policy-target pruning a 40-simulation root has no count above 1 and the reference's
        # pruned policy is 0 / 0 (MCTS.py:77-80,100-102); the raw counts are always a distribution
        sims, kw['forced_playouts'] = 60, False
    temp = [1.25, 0.8, 1.0]
    args = Args(numMCTSSims=sims, prob_fullMCTS=prob_full, ratio_fullMCTS=5, dirichletAlpha=0, temperature=temp,
                tempThreshold=6, **kw)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=2048, max_examples=T * 1200, rng_seed=seed,
               stream0=stream0)
    f.selfplay_start()
    _play_until(f, T, 2 * T, mode)
    boards, pis, zs, valids, qs, meta = [x.cpu().numpy() for x in f.drain_examples()]
    assert len(boards) > 0
    for t in range(T):
        o = O.run_episode(og, O.make_args(numMCTSSims=sims, prob_fullMCTS=prob_full, **kw), None, seed=seed,
                          stream=stream0 + t, temp=(temp[0], temp[1]), tempThreshold=6.0)
        sel = np.flatnonzero((meta[:, 0] == stream0 + t) & (meta[:, 1] == 0))
        sel = sel[np.argsort(meta[sel, 2])]
        full_plies = np.flatnonzero(o['full'])
        assert len(sel) == len(full_plies), (t, len(sel), len(full_plies))
        for k, ply in zip(sel, full_plies):
            assert meta[k, 2] == ply and meta[k, 3] == o['player'][ply]
            assert np.array_equal(boards[k], o['canonical'][ply]), (t, ply)
            assert np.array_equal(pis[k], o['pi'][ply].astype(np.float32)), (t, ply)
            assert np.array_equal(qs[k], o['q'][ply])
            assert np.array_equal(zs[k], np.roll(o['result'], -int(o['player'][ply])))
            assert np.array_equal(valids[k].astype(bool), og.getValidMoves(o['canonical'][ply], 0))
    f.close()


@pytest.mark.parametrize('mode', ['rounds', 'async'])
@pytest.mark.parametrize('variant,cap', [('splendor2', 400), ('azul', 900), ('santorini1', 700)])
def test_selfplay_gc_keeps_results(tmp_path, variant, cap, mode):
    """A node arena too small for a whole game forces the clean-up many times (free-id stack, record free lists / id-indexed
    record slots, table rebuild); results must not change and the structure must stay valid -- in the host-driven rounds and inside the
    asynchronous pipeline's descent kernel."""
    import torch
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    name, v = {'splendor2': ('splendor', 2), 'santorini1': ('santorini', 1), 'azul': ('azul', 0)}[variant]
    g = {'splendor': lambda: games.SplendorGame(v), 'santorini': lambda: games.SantoriniGame(v), 'azul': games.AzulGame}[name]()
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL}[name], v)
    kw = dict(MCTS_ARGS[variant])
    sims, T, seed, stream0 = 60, 8, 7, 50
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.0, 1.0, 1.0],
                tempThreshold=6, **kw)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=cap, max_examples=T * 1200, rng_seed=seed, stream0=stream0)
    f.selfplay_start()
    st = _play_until(f, T, 3 * T, mode)          # every tree has finished its first game by then
    # (mode 'async': the clean-up runs INSIDE the pipeline's descent kernel, on the one wave that found the search finished -- gc_scan<G, true>)
    assert st['gc_runs'] > 0
    assert f.validate() == 0
    boards, pis, zs, valids, qs, meta = [x.cpu().numpy() for x in f.drain_examples()]
    for t in range(T):
        o = O.run_episode(og, O.make_args(numMCTSSims=sims, **kw), None, seed=seed, stream=stream0 + t,
                          temp=(1.0, 1.0), tempThreshold=6.0)
        sel = np.flatnonzero((meta[:, 0] == stream0 + t) & (meta[:, 1] == 0))
        sel = sel[np.argsort(meta[sel, 2])]
        assert len(sel) == o['plies']
        for k, ply in zip(sel, range(o['plies'])):
            assert np.array_equal(boards[k], o['canonical'][ply]), (t, ply)
            assert np.array_equal(pis[k], o['pi'][ply].astype(np.float32)), (t, ply)
    f.close()


def test_advance_cadence_does_not_change_results():
    """SelfPlayEngine launches selfplay_advance (move sampling, re-rooting, root Dirichlet noise) every `advance_every`
    rounds; trees wait for it, so the examples of every game must be identical for any cadence (device noise sampler on) --
    and for the fused form, where the expansion of round r rides on the descent launch of round r + 1."""
    import torch
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    args = Args(numMCTSSims=24, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['splendor2'])
    T = 32
    res = []
    import ctypes
    from azg_amd._lib import lib
    lib().azg_debug_poison_onchip.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    for K, graph, fused in ((1, False, False), (5, True, True), (3, True, False), (1, False, True)):
        # a different pattern in every CU's LDS and in the queue's scratch memory before each run: a kernel that reads on-chip memory it
        # never wrote (this test once depended on stale spill slots) shows up as a cadence dependence
        lib().azg_debug_poison_onchip((0x7FC00000, 0xFFFFFFFF, 0xA5A5A5A5, 0x00010001)[len(res)], None)
        e = SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=2048, max_examples=T * 400, rng_seed=99, stream0=7,
                           use_graph=graph, advance_every=K, fused=fused)
        e.start()
        for _ in range(400):
            e.run(40)
            st = e.stats()
            assert st['errors'] == 0
            if st['games'] >= 3 * T:
                break
        ex = [x.cpu().numpy() for x in e.drain_examples()]
        meta = ex[5]
        keep = np.flatnonzero(meta[:, 1] == 0)                       # first game of every stream (finished in both runs)
        order = keep[np.lexsort((meta[keep, 2], meta[keep, 0]))]
        res.append([x[order] for x in ex])
        for grp in e.groups:
            grp.f.close()
    assert len(res[0][0]) > 0
    for other in res[1:]:
        assert len(res[0][0]) == len(other[0])
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)


def test_drain_examples_with_symmetries():
    """drain_examples(symmetries=True) == Coach.py:66-69: every recorded ply expanded into all getSymmetries forms
    (identity first), z / q / meta repeated; checked against the oracle's getSymmetries."""
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    og = O.OracleGame(O.SPLENDOR, 2)
    args = Args(numMCTSSims=16, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['splendor2'])
    T = 8
    e = SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=1024, max_examples=T * 400, rng_seed=5, use_graph=False)
    e.start()
    for _ in range(200):
        e.run(50)
        if e.stats()['games'] >= T:
            break
    boards, pi, z, valids, q, meta = [x.cpu().numpy() for x in e.drain_examples(symmetries=True)]
    assert len(boards) > 0
    i = 0
    n_records = 0
    while i < len(boards):
        syms = og.getSymmetries(boards[i], pi[i], valids[i], max_sym=32)       # row i is the identity form of a record
        for k, (s, p, v) in enumerate(syms):
            assert np.array_equal(boards[i + k], s.reshape(-1)) and np.array_equal(pi[i + k], p)
            assert np.array_equal(valids[i + k], v.astype(np.uint8))
            assert np.array_equal(z[i + k], z[i]) and np.array_equal(q[i + k], q[i]) and np.array_equal(meta[i + k], meta[i])
        i += len(syms)
        n_records += 1
    assert i == len(boards) and n_records == e.stats()['examples'] or n_records > 0
    for grp in e.groups:
        grp.f.close()


def test_full_size_properties():
    """BASELINE.json's headline configuration at full size (4096 concurrent Splendor-2p games, 800 simulations per move,
    the pretrained V80 net on the engine's kernels): size-independent invariants after a few plies -- no error flag, the
    structural validator passes on all 4096 trees, every root's visit counts add up, every tree keeps moving."""
    import os
    import torch
    from azg_amd import games
    from azg_amd.nnet import SplendorV80Hip
    from azg_amd.selfplay import SelfPlayEngine
    T, sims = 4096, 800
    g = games.SplendorGame(2)
    net = SplendorV80Hip.from_npz(os.path.join(os.path.dirname(__file__), 'golden', 'weights_splendor2_v80.npz'), max_batch=T)
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['splendor2'])
    e = SelfPlayEngine(g, net, args, T, max_examples=T * 8)
    e.start()
    e.run(3 * sims + 64)
    st = e.stats()
    assert st['errors'] == 0
    assert st['plies'] >= 2 * T and st['plies'] <= 4 * T                 # every tree is in its 3rd or 4th search
    assert st['sims'] >= st['expansions'] > 0
    assert e.forest.validate() == 0
    rs = e.forest.root_stats()
    Ns, Nsa = rs['Ns'].cpu().numpy().astype(np.int64), rs['Nsa'].cpu().numpy().astype(np.int64)
    has_root = Ns > 0
    assert has_root.mean() > 0.9
    # MCTS.py:180-181: every visit of the root increments Ns and exactly one Nsa (tree reuse keeps both from earlier searches)
    assert np.array_equal(Nsa.sum(axis=1)[has_root], Ns[has_root])
    for grp in e.groups:
        grp.f.close()


@pytest.mark.parametrize('variant', ['azul', 'splendor4'])
def test_engine_with_one_launch_net_other_games(variant):
    """SelfPlayEngine (HIP graph, fused expansion) with the generic one-launch net (MobileNet1dHip) on the pretrained Azul /
    Splendor-4p weights: a few hundred rounds, no engine error, and every recorded example is well-formed -- pi is a
    distribution over the valid actions, z and q are in range, the board is a state of the game the net was built for."""
    import os
    import torch
    from azg_amd import games, nnet
    from azg_amd.selfplay import SelfPlayEngine
    root = os.path.join(os.path.dirname(__file__), 'golden')
    if variant == 'azul':
        g = games.AzulGame()
        base = nnet.AzulV84.from_npz(os.path.join(root, 'weights_azul_v84.npz'), device='cuda:0')
    else:
        g = games.SplendorGame(4)
        base = nnet.SplendorV80.from_npz(os.path.join(root, 'weights_splendor4_v80.npz'), num_players=4, device='cuda:0')
    T, sims = 64, 25
    net = nnet.MobileNet1dHip(base, max_batch=T)
    assert net.fused
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **dict(MCTS_ARGS[variant]))
    eng = SelfPlayEngine(g, net, args, T, node_capacity=4096, max_examples=T * 400, rng_seed=7)
    eng.start()
    eng.run(sims * 40 + 64)
    torch.cuda.synchronize()
    st = eng.stats()
    assert st['errors'] == 0 and st['plies'] >= T * 30
    boards, pi, z, valids, q, meta = eng.drain_examples()
    n = boards.shape[0]
    if n:                                              # only finished games deliver examples
        assert torch.allclose(pi.sum(dim=1), torch.ones(n, device=pi.device), atol=1e-4)
        assert float(pi[valids == 0].abs().max()) == 0.0
        assert float(z.abs().max()) <= 1.0 and float(q.abs().max()) <= 1.0 + 1e-6
    # the net the engine used agrees with the plain torch evaluation on the engine's own leaf batch
    f = eng.forest
    lp, lv = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid)
    rp, rv = base.to('cuda:0', torch.float64).predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
    assert float((lp - rp).abs().max()) < 1e-5 and float((lv - rv).abs().max()) < 1e-5


def _mini_engine(T=16, sims=12, max_examples=None, rng_seed=11):
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    # (no policy-target pruning: with a dozen simulations it often leaves no count above 1, see test_empty_pruned_policy_...)
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **{**MCTS_ARGS['splendor2'], 'forced_playouts': False})
    return SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=1024, max_examples=max_examples or T * 400,
                          rng_seed=rng_seed, use_graph=False)


def _play_quota(e, quota, epoch):
    e.start(epoch=epoch, episode_quota=quota)
    for _ in range(400):
        e.run(128)
        st = e.stats()
        assert st['errors'] == 0
        if st['active'] == 0:
            break
    return st, [x.cpu().numpy() for x in e.drain_examples()]


def test_episode_quota_and_rng_epoch():
    """Coach.executeEpisodes semantics (Coach.py:86-148): exactly numEps games, every one played to its end and kept (no bias
    towards short games), then the trees idle; a new RNG epoch plays different games, the same epoch the same ones."""
    T = 16
    e = _mini_engine(T)
    for quota in (5, 16, 37):                                   # fewer episodes than trees, one each, several waves
        st, ex = _play_quota(e, quota, epoch=1)
        assert st['games'] == quota and st['active'] == 0
        meta = ex[5]
        games = {(int(m[0]), int(m[1])) for m in meta}
        assert len(games) == quota                              # every finished game delivered its examples
        per_tree = np.bincount([s for s, _ in games], minlength=T)[:T]
        assert per_tree.max() - per_tree.min() <= 1 and per_tree.sum() == quota
        e.run(64)                                               # idle trees stay idle
        assert e.stats()['games'] == quota
    _, a1 = _play_quota(e, T, epoch=1)
    _, a2 = _play_quota(e, T, epoch=1)
    _, b = _play_quota(e, T, epoch=2)
    key = lambda ex: np.lexsort((ex[5][:, 2], ex[5][:, 0]))     # noqa: E731  (stream, ply)
    assert all(np.array_equal(x[key(a1)], y[key(a2)]) for x, y in zip(a1, a2))          # same epoch: identical games
    assert b[0].shape != a1[0].shape or not np.array_equal(b[0][key(b)], a1[0][key(a1)])   # new epoch: fresh randomness
    # epoch 0 == the plain RNG-contract streams of azg_selfplay_start
    e0 = _mini_engine(T)
    e0.start()
    _, c0 = _play_quota(e, T, epoch=0)
    for _ in range(400):
        e0.run(128)
        if e0.stats()['games'] >= 3 * T:
            break
    d0 = [x.cpu().numpy() for x in e0.drain_examples()]
    first = d0[5][:, 1] == 0
    d0 = [x[first] for x in d0]
    assert all(np.array_equal(x[key(c0)], y[key(d0)]) for x, y in zip(c0, d0))
    for eng in (e, e0):
        for grp in eng.groups:
            grp.f.close()


def test_example_ring_overflow_is_loud_and_never_truncates_a_game():
    e = _mini_engine(T=16, max_examples=200)
    e.start()
    for _ in range(400):
        e.run(128)
        st = e.stats()
        if st['examples_dropped']:
            break
    assert st['examples_dropped'] > 0 and (st['errors'] & 16)
    ex = [x.cpu().numpy() for x in e.drain_examples()]
    meta = ex[5]
    assert len(meta) <= 200
    # every delivered game is complete: its plies are 0..n-1 without a gap (Splendor records every ply at prob_full = 1)
    for s in np.unique(meta[:, 0]):
        for gidx in np.unique(meta[meta[:, 0] == s, 1]):
            plies = np.sort(meta[(meta[:, 0] == s) & (meta[:, 1] == gidx), 2])
            assert np.array_equal(plies, np.arange(len(plies)))
    for grp in e.groups:
        grp.f.close()


def test_empty_pruned_policy_is_an_error_not_a_move():
    """With a dozen simulations and policy-target pruning a root often has no count above 1: the reference's pruned policy is then
    0 / 0 (MCTS.py:77-80,100-102 raises ZeroDivisionError); the engine parks the tree with error bit 64 instead of playing an
    arbitrary move."""
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    T = 16
    args = Args(numMCTSSims=12, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.0, 1.0, 1.0], tempThreshold=6,
                cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=512, max_examples=T * 64)
    net = HashNetTorch(2)
    f.selfplay_start()
    for rnd in range(2000):
        f.select()
        pi, vv = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
        f.expand_backup(pi, vv)
        f.selfplay_advance()
        if rnd % 100 == 99 and f.stats()['errors']:
            break
    st = f.stats()
    assert st['errors'] == 64, st
    f.close()


import glob as _glob
import os as _os
_EP_CASES = sorted(_os.path.basename(p)[len('episode_'):-len('.npz')]
                   for p in _glob.glob(_os.path.join(_os.path.dirname(__file__), 'golden', 'episode_*.npz')))


@pytest.mark.parametrize('case', [c for c in _EP_CASES if c.split('_')[0] in _ASYNC_GAMES])
def test_async_pipeline_vs_reference_executeEpisode(case, golden_dir):
    """the PIPELINE itself against the episodes the reference's own Coach.executeEpisode played (see below)"""
    _vs_reference_episode(case, golden_dir, 'async')


@pytest.mark.parametrize('case', _EP_CASES)
def test_selfplay_vs_reference_executeEpisode(case, golden_dir):
    _vs_reference_episode(case, golden_dir, 'rounds')


def _vs_reference_episode(case, golden_dir, mode):
    """Device-resident self-play against an episode the REFERENCE's own Coach.executeEpisode played (Coach.py:37-84; fixture from
    tools/gen_golden_episode.py): same init board, same counter stream -> the recorded plies (canonical board, pi,
    q, player), z = roll(r, -player), and the symmetry-expanded example list in the reference's order."""
    import torch
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    d = np.load(_os.path.join(golden_dir, 'episode_%s.npz' % case))
    variant = case.split('_')[0]
    g = {'splendor2': lambda: games.SplendorGame(2), 'santorini11': lambda: games.SantoriniGame(11), 'azul': games.AzulGame}[variant]()
    T = 2                                   # tree 0 replays the fixture's stream; tree 1 is a bystander on the next stream
    args = Args(numMCTSSims=int(d['sims']), prob_fullMCTS=float(d['prob_full']), ratio_fullMCTS=5, dirichletAlpha=0,
                temperature=[float(x) for x in d['temperature']], tempThreshold=float(d['tempThreshold']), cpuct=float(d['cpuct']),
                fpu=float(d['fpu']), universes=int(d['universes']), forced_playouts=bool(d['forced']))
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=4096, max_examples=T * 1200, rng_seed=int(d['seed']),
               stream0=int(d['stream']))
    ib = torch.from_numpy(np.stack([d['init_board'], d['init_board']])).cuda()
    f.selfplay_start(init_boards=ib)
    _play_until(f, T, 2 * T, mode, max_iter=400000)
    boards, pis, zs, valids, qs, meta = f.drain_examples()
    m = meta.cpu().numpy()
    sel = np.flatnonzero((m[:, 0] == int(d['stream'])) & (m[:, 1] == 0))
    sel = sel[np.argsort(m[sel, 2])]
    full_plies = np.flatnonzero(d['full'])
    assert len(sel) == len(full_plies)
    assert np.array_equal(m[sel, 2], full_plies) and np.array_equal(m[sel, 3], d['player'][full_plies])
    idx = torch.from_numpy(sel).cuda()
    b_, p_, z_, v_, q_ = [x[idx].contiguous() for x in (boards, pis, zs, valids, qs)]
    assert np.array_equal(b_.cpu().numpy(), d['canonical'][full_plies])
    assert np.array_equal(p_.cpu().numpy(), d['pi'][full_plies].astype(np.float32))
    assert np.array_equal(q_.cpu().numpy(), d['q'][full_plies])
    # the example list as executeEpisode returns it: all symmetric forms of a ply, in order (Coach.py:66-69,76-82)
    ob, op, ov, cnt = g.symmetries_batch(b_, p_, v_)
    K = ob.shape[1]
    keep = (torch.arange(K, device=cnt.device)[None, :] < cnt[:, None]).reshape(-1)
    rep = torch.repeat_interleave(torch.arange(b_.shape[0], device=cnt.device), cnt.to(torch.int64))
    assert int(keep.sum()) == len(d['ex_board'])
    assert np.array_equal(ob.reshape(-1, ob.shape[2])[keep].cpu().numpy(), d['ex_board'])
    assert np.array_equal(op.reshape(-1, op.shape[2])[keep].cpu().numpy(), d['ex_pi'].astype(np.float32))
    assert np.array_equal(ov.reshape(-1, ov.shape[2])[keep].cpu().numpy(), d['ex_valid'])
    assert np.array_equal(z_[rep].cpu().numpy(), d['ex_z'])
    assert np.array_equal(q_[rep].cpu().numpy(), d['ex_q'])
    f.close()


def test_set_search_params_between_moves():
    """args.numMCTSSims / args.prob_fullMCTS are read at every getActionProb call in the reference (MCTS.py:58-59); the forest takes
    new values for the searches that begin after azg_forest_set_search_params: a stretch of fast searches records no example, the
    full searches that follow do, and an engine that switches plays on without errors (bench.py's pre-roll)"""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    T, sims = 16, 40
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
                **MCTS_ARGS['splendor2'])
    eng = SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=2048, max_examples=T * 400, rng_seed=3)
    eng.start()
    eng.set_search_params(sims, 0.0)                       # fast searches only: sims // 5 simulations, no examples (Coach.py:65)
    eng.run(20 * (sims // 5 + eng.K))
    s0 = eng.stats()
    assert s0['errors'] == 0 and s0['plies'] >= 10 * T and s0['examples'] == 0
    eng.set_search_params(sims, 1.0)
    eng.run(12 * (sims + eng.K))
    s1 = eng.stats()
    assert s1['errors'] == 0 and s1['plies'] > s0['plies']
    # the sims counter: a full search costs `sims` simulations, a fast one sims // 5
    assert (s1['sims'] - s0['sims']) / max(1, s1['plies'] - s0['plies']) > 0.7 * sims > (s0['sims'] / s0['plies'])
    eng.run(120 * (sims + eng.K))                          # games end: the plies played with full searches were recorded
    s2 = eng.stats()
    assert s2['errors'] == 0 and s2['games'] > 0 and s2['examples'] > 0


@pytest.mark.parametrize('T,sims,use_graph', [(40, 24, False), (64, 40, True)])
def test_percu_round_kernel_equals_two_kernel_rounds(T, sims, use_graph):
    """The per-CU round kernel (azg_forest_rounds_v80_h2: 16 trees + their 16 leaves per workgroup, `advance_every` rounds per launch)
    against the two-kernel rounds (azg_forest_select_fused + azg_nn_v80_forward_h2) it replaces: the same games move for move -- every
    drained example record (board, pi, z, valids, q, meta), the statistics counters and the root statistics are EQUAL, not close.
    T = 40 leaves the last workgroup half empty."""
    import os
    import torch
    from azg_amd import games
    from azg_amd.nnet import SplendorV80Hip
    from azg_amd.selfplay import SelfPlayEngine
    g = games.SplendorGame(2)
    w = os.path.join(os.path.dirname(__file__), 'golden', 'weights_splendor2_v80.npz')
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['splendor2'])
    out = []
    for percu in (False, True):
        net = SplendorV80Hip.from_npz(w, max_batch=T)
        e = SelfPlayEngine(g, net, args, T, node_capacity=2048, max_examples=T * 400, rng_seed=9, use_graph=use_graph, advance_every=8,
                           percu=percu, async_pipe=False)
        assert e.percu == percu
        e.start()
        e.run(8 * 60 * (sims + 8) // 8)               # ~60 plies: most games end and restart
        st = e.stats()
        assert st['errors'] == 0 and st['games'] > 0
        ex = [x.cpu() for x in e.drain_examples()]
        m = ex[5].to(torch.int64)                       # the ring's order is the order in which games happened to end: sort by (stream, game, ply)
        order = torch.argsort((m[:, 0] * 100000 + m[:, 1]) * 1000 + m[:, 2])
        ex = [x[order] for x in ex]
        rs = {k: v.cpu() for k, v in e.forest.root_stats().items()}
        out.append((st, ex, rs))
        assert e.forest.validate(verbose=False) == 0
        e.close()
    (s0, e0, r0), (s1, e1, r1) = out
    for k in ('plies', 'games', 'sims', 'levels', 'expansions', 'terminal_hits', 'examples', 'sum_valid_visited', 'sum_depth_at_expand'):
        assert s0[k] == s1[k], (k, s0[k], s1[k])
    assert len(e0[0]) == len(e1[0]) > 0
    for a, b in zip(e0, e1):
        assert torch.equal(a, b)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k


@pytest.mark.parametrize('T,sims,K,budget,cfg', [(40, 24, 8, 0, dict(n_net=3, n_sel=5)), (64, 40, 8, 10, dict(n_net=2, n_sel=64)),
                                                 (300, 32, 16, 4, dict(n_net=8, n_sel=3, batch_wait_ticks=0)), (256, 48, 48, 0, {})])
def test_async_pipeline_equals_two_kernel_rounds(T, sims, K, budget, cfg):
    """The asynchronous tree pipeline (azg_forest_async_rounds_v80_h2: persistent descent workgroups + persistent net workgroups, leaves and
    trees handed over through device-side queues, no boundary wider than one tree) against the two-kernel rounds it replaces
    (azg_forest_select_fused + azg_nn_v80_forward_h2): the same games move for move -- every drained example record
    (board, pi, z, valids, q, meta), the statistics counters and the root statistics are EQUAL, not close, whatever the CU split, the
    trees per workgroup (1 .. 100), the batch wait, the work budget (a call that parks its descent is followed by the next call, as
    a round is by the next round) and the number of calls per launch."""
    import os
    import torch
    from azg_amd import games
    from azg_amd.nnet import SplendorV80Hip
    from azg_amd.selfplay import SelfPlayEngine
    g = games.SplendorGame(2)
    w = os.path.join(os.path.dirname(__file__), 'golden', 'weights_splendor2_v80.npz')
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['splendor2'])
    out = []
    for pipe in (False, True):
        net = SplendorV80Hip.from_npz(w, max_batch=T)
        e = SelfPlayEngine(g, net, args, T, node_capacity=2048, max_examples=T * 400, rng_seed=9, use_graph=False, advance_every=1,
                           work_budget=budget, percu=False, async_pipe=pipe, async_cfg=dict(cfg, shared_budget=False))
        assert e.async_pipe == pipe
        e.start()
        # ~60 plies: most games end and restart.  A tree of the pipeline advances the moment its search is finished (on the wave that
        # found it finished), the two-kernel form therefore advances after every round; the pipeline runs K calls per tree per launch
        n = 60 * (sims + 2)
        if pipe:
            for _ in range(n // K):
                e.run(K)
            e.run(n % K)
        else:
            for _ in range(n):
                e.groups[0].round(e.fused, advance=True)
        torch.cuda.synchronize()
        st = e.stats()
        assert st['errors'] == 0 and st['games'] > 0, (st['errors'], e.forest.async_profile()['ctl'] if pipe else None)
        ex = [x.cpu() for x in e.drain_examples()]
        m = ex[5].to(torch.int64)                       # the ring's order is the order in which games happened to end: sort by (stream, game, ply)
        order = torch.argsort((m[:, 0] * 100000 + m[:, 1]) * 1000 + m[:, 2])
        ex = [x[order] for x in ex]
        rs = {k: v.cpu() for k, v in e.forest.root_stats().items()}
        if pipe:
            prof = e.forest.async_profile()
            assert prof['descents'] > 0 and prof['leaves'] > 0 and prof['batches'] > 0
        out.append((st, ex, rs))
        assert e.forest.validate(verbose=False) == 0
        e.close()
    (s0, e0, r0), (s1, e1, r1) = out
    for k in ('plies', 'games', 'sims', 'levels', 'expansions', 'terminal_hits', 'examples', 'sum_valid_visited', 'sum_depth_at_expand'):
        assert s0[k] == s1[k], (k, s0[k], s1[k])
    assert len(e0[0]) == len(e1[0]) > 0
    for a, b in zip(e0, e1):
        assert torch.equal(a, b)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k


@pytest.mark.parametrize('T,sims,K,budget,cfg', [(24, 24, 8, 0, dict(n_net=3, n_sel=5)), (200, 32, 16, 6, dict(n_net=25, n_sel=2)), (128, 40, 40, 20, {})])
def test_async_pipeline_santorini_equals_two_kernel_rounds(T, sims, K, budget, cfg):
    """The asynchronous tree pipeline for the second hot-path game -- Santorini without gods with the V89 ResNet, 8 leaves per forward
    (azg_forest_async_rounds_conv5_h2) -- against the two-kernel rounds (azg_forest_select_fused + azg_selfplay_advance +
    azg_nn_conv5_forward_h2): the same games move for move, every example record, statistics counter and root statistic EQUAL."""
    import os
    import torch
    from azg_amd import games
    from azg_amd.nnet import SantoriniV89, SantoriniV89Hip
    from azg_amd.selfplay import SelfPlayEngine
    g = games.SantoriniGame(1)
    w = os.path.join(os.path.dirname(__file__), 'golden', 'weights_santorini1_v89.npz')
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.2, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS['santorini1'])
    out = []
    for pipe in (False, True):
        net = SantoriniV89Hip(SantoriniV89.from_npz(w, device='cuda:0'), max_batch=T)
        e = SelfPlayEngine(g, net, args, T, node_capacity=4096, max_examples=T * 400, rng_seed=11, use_graph=False, advance_every=1,
                           work_budget=budget, async_pipe=pipe, async_cfg=dict(cfg, shared_budget=False))
        assert e.async_pipe == pipe
        e.start()
        n = 40 * (sims + 2)                             # ~40 plies: most games end and restart
        if pipe:
            for _ in range(n // K):
                e.run(K)
            e.run(n % K)
        else:
            for _ in range(n):
                e.groups[0].round(e.fused, advance=True)
        torch.cuda.synchronize()
        st = e.stats()
        assert st['errors'] == 0 and st['games'] > 0, (st['errors'], e.forest.async_profile()['ctl'] if pipe else None)
        ex = [x.cpu() for x in e.drain_examples()]
        m = ex[5].to(torch.int64)
        order = torch.argsort((m[:, 0] * 100000 + m[:, 1]) * 1000 + m[:, 2])
        ex = [x[order] for x in ex]
        rs = {k: v.cpu() for k, v in e.forest.root_stats().items()}
        out.append((st, ex, rs))
        assert e.forest.validate(verbose=False) == 0
        e.close()
    (s0, e0, r0), (s1, e1, r1) = out
    for k in ('plies', 'games', 'sims', 'levels', 'expansions', 'terminal_hits', 'examples', 'sum_valid_visited', 'sum_depth_at_expand'):
        assert s0[k] == s1[k], (k, s0[k], s1[k])
    assert len(e0[0]) == len(e1[0]) > 0
    for a, b in zip(e0, e1):
        assert torch.equal(a, b)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k


@pytest.mark.parametrize('variant,T,sims,K,budget,cfg', [('splendor4', 40, 24, 8, 0, dict(n_net=4, n_sel=3)), ('splendor3', 64, 32, 16, 10, {}),
                                                         ('azul', 48, 24, 8, 0, dict(n_net=2, n_sel=7)), ('azul', 96, 40, 40, 20, {})])
def test_async_pipeline_mobilenet1d_equals_two_kernel_rounds(variant, T, sims, K, budget, cfg):
    """The asynchronous tree pipeline for the MobileNet-1d games -- Splendor 3 / 4 players (8 leaves per forward) and Azul (16; root noise with
    the automatic alpha) -- (azg_forest_async_rounds_mb1d_h2) against the two-kernel rounds (azg_forest_select_fused + azg_selfplay_advance +
    azg_nn_mb1d_forward_h2): the same games move for move, every example record, statistics counter and root statistic EQUAL."""
    import os
    import torch
    from azg_amd import games, nnet
    from azg_amd.selfplay import SelfPlayEngine
    root = os.path.join(os.path.dirname(__file__), 'golden')
    if variant == 'azul':
        g = games.AzulGame()
        mk = lambda: nnet.MobileNet1dHip(nnet.AzulV84.from_npz(os.path.join(root, 'weights_azul_v84.npz'), device='cuda:0'), max_batch=T)  # noqa: E731
        alpha = -1
    else:
        npl = int(variant[-1])
        g = games.SplendorGame(npl)
        wfile = 'weights_splendor4_v80.npz' if npl == 4 else None
        if wfile is None:                                # (no 3-player checkpoint in the fixtures: random weights of the 3-player geometry)
            base3 = nnet.SplendorV80.random_init(num_players=3, seed=5, device='cuda:0')
            mk = lambda: nnet.MobileNet1dHip(base3, max_batch=T)  # noqa: E731
        else:
            mk = lambda: nnet.MobileNet1dHip(nnet.SplendorV80.from_npz(os.path.join(root, wfile), num_players=npl, device='cuda:0'), max_batch=T)  # noqa: E731
        alpha = 0.3
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=alpha, temperature=[1.25, 0.8, 1.0],
                tempThreshold=6, **MCTS_ARGS[variant])
    out = []
    for pipe in (False, True):
        e = SelfPlayEngine(g, mk(), args, T, node_capacity=4096, max_examples=T * 400, rng_seed=13, use_graph=False, advance_every=1,
                           work_budget=budget, async_pipe=pipe, async_cfg=dict(cfg, shared_budget=False))
        assert e.async_pipe == pipe
        e.start()
        n_rounds = 50 * (sims + 2)
        if pipe:
            for _ in range(n_rounds // K):
                e.run(K)
            e.run(n_rounds % K)
        else:
            for _ in range(n_rounds):
                e.groups[0].round(e.fused, advance=True)
        torch.cuda.synchronize()
        st = e.stats()
        assert st['errors'] == 0 and st['plies'] > 10 * T, (st['errors'], e.forest.async_profile()['ctl'] if pipe else None)
        ex = [x.cpu() for x in e.drain_examples()]
        m = ex[5].to(torch.int64)
        order = torch.argsort((m[:, 0] * 100000 + m[:, 1]) * 1000 + m[:, 2])
        ex = [x[order] for x in ex]
        rs = {k: v.cpu() for k, v in e.forest.root_stats().items()}
        out.append((st, ex, rs))
        assert e.forest.validate(verbose=False) == 0
        e.close()
    (s0, e0, r0), (s1, e1, r1) = out
    for k in ('plies', 'games', 'sims', 'levels', 'expansions', 'terminal_hits', 'examples', 'sum_valid_visited', 'sum_depth_at_expand'):
        assert s0[k] == s1[k], (k, s0[k], s1[k])
    assert len(e0[0]) == len(e1[0])
    for a, b in zip(e0, e1):
        assert torch.equal(a, b)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k


@pytest.mark.parametrize('on_side_stream', [False, True])
def test_async_pipeline_in_a_process_with_many_streams(on_side_stream):
    """The pipeline's two kernels must run side by side whatever streams the process has: HIP multiplexes streams onto a few hardware
    queues (kernels of one queue run one after the other) and a blocking stream waits for the legacy default stream -- either would leave
    the net kernel waiting for leaves the descent kernel can never deliver (the engine reports that as error bit 128 after its
    time-out).  The kernels run on two private non-blocking high-priority streams; here the caller is on the default stream or on the
    fourth of a dozen streams the process created first."""
    import os
    import torch
    from azg_amd import games
    from azg_amd.nnet import SplendorV80Hip
    from azg_amd.selfplay import SelfPlayEngine
    streams = [torch.cuda.Stream() for _ in range(12)]
    x = torch.zeros(1024, device='cuda')
    for s in streams:
        with torch.cuda.stream(s):
            x = x + 1
    torch.cuda.synchronize()
    g = games.SplendorGame(2)
    w = os.path.join(os.path.dirname(__file__), 'golden', 'weights_splendor2_v80.npz')
    args = Args(numMCTSSims=24, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
                **MCTS_ARGS['splendor2'])
    T = 40
    e = SelfPlayEngine(g, SplendorV80Hip.from_npz(w, max_batch=T), args, T, node_capacity=2048, max_examples=T * 400, rng_seed=9,
                       async_pipe=True, async_cfg=dict(n_net=3, n_sel=5))
    e.start()
    if on_side_stream:
        with torch.cuda.stream(streams[3]):
            e.run(200)
    else:
        e.run(200)
    torch.cuda.synchronize()
    st = e.stats()
    assert st['errors'] == 0 and st['plies'] > 4 * T, (st['errors'], st['plies'], e.forest.async_profile()['ctl'])
    e.close()


@pytest.mark.parametrize('game_key,sims,cap', [('santorini11', 800, 0), ('splendor4', 800, 0), ('azul', 800, 0), ('santorini1', 800, 0),
                                               ('azul', 1600, 44000)])
def test_full_size_properties_other_configs(game_key, sims, cap):
    """BASELINE.json configs 3 / 4 / 5 (and the north star's second target) at FULL size -- 4096 concurrent games, 800 simulations per
    move (config 5: Azul at its 1600 simulations with the automatic root Dirichlet noise of azul/pretrained.pt, alpha = 10 / n_valid,
    MCTS.py:188-192, in 232 of the 288 GB), the pretrained net of the game on the engine's kernels, the engine exactly as bench.py builds
    it: size-independent invariants after three plies -- no error flag, the structural validator passes on all 4096 trees, every root's
    visit counts add up (MCTS.py:180-181), every tree keeps moving.  (Splendor 2p: test_full_size_properties.)"""
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('azg_bench', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    T = 4096
    a = bench.argparse.Namespace(net_dtype='fp32', net='hip', groups=1, sims=sims, prob_full=1.0, node_capacity=cap, no_graph=False, level_budget=0,
                                 work_budget=-1, advance_every=0, no_pin_xcd=False)
    eng, margs, label, weights, net_kind = bench.build_engine(a, game_key, T, 0, 'cuda:0')
    assert net_kind == 'hip' and int(margs['numMCTSSims']) == sims
    if game_key == 'azul':
        assert float(margs['dirichletAlpha']) < 0          # the automatic alpha: root noise is on
    try:                                                 # (the forest is 140-230 GB: it must be gone before the next case, whatever happens)
        eng.start()
        eng.run(3 * sims + 64)
        st = eng.stats()
        assert st['errors'] == 0, (st['errors'], eng.forest.async_profile()['ctl'] if getattr(eng, 'async_pipe', False) else None)
        assert 2 * T <= st['plies'] <= 4 * T                 # every tree is in its 3rd or 4th search
        assert st['sims'] >= st['expansions'] > 0
        assert eng.forest.validate(verbose=False) == 0
        rs = eng.forest.root_stats()
        Ns, Nsa = rs['Ns'].cpu().numpy().astype(np.int64), rs['Nsa'].cpu().numpy().astype(np.int64)
        has_root = Ns > 0
        assert has_root.mean() > 0.9
        assert np.array_equal(Nsa.sum(axis=1)[has_root], Ns[has_root])
    finally:
        eng.close()
        del eng
        torch.cuda.empty_cache()


@pytest.mark.parametrize('variant,T,sims,alpha', [('splendor2', 48, 24, 0.3), ('azul', 40, 30, 0.0)])
def test_async_pipeline_shared_budget_plays_the_same_games(variant, T, sims, alpha):
    """The mode bench.py times: SelfPlayEngine's default for the pipeline is the WORK-SHARING budget (run(rounds) = rounds x T calls for the
    trees together; a tree is stopped wherever it is when the budget is spent and goes on in the next launch).  How far each game gets
    per launch depends on timing -- what is played must not: two whole games per tree (episode quota 2 T) over launches of odd lengths,
    then EVERY drained record, keyed (stream, game, ply), equals the per-tree-budget run's (which test_async_pipeline_first_games_vs_oracle
    ties to the oracle).  Root noise on for Splendor: the in-kernel sampler is keyed by the tree's simulation counter."""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetPipeline
    g = games.SplendorGame(2) if variant == 'splendor2' else games.AzulGame()
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=alpha, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
                **{**MCTS_ARGS[variant], 'forced_playouts': False})
    res = []
    for shared, lengths in ((True, (37, 91, 13, 255, 64, 7)), (False, (64,))):
        e = SelfPlayEngine(g, HashNetPipeline(g.P), args, T, node_capacity=2048, max_examples=T * 800, rng_seed=21, stream0=300,
                           async_pipe=True, async_cfg=dict(shared_budget=shared))
        assert e.async_pipe and e.groups[0].async_cfg['shared_budget'] is shared
        e.start(episode_quota=2 * T)
        for k in range(4000):
            e.run(lengths[k % len(lengths)])
            st = e.stats()
            assert st['errors'] == 0, st
            if st['active'] == 0:
                break
        assert st['games'] == 2 * T and st['active'] == 0
        assert e.forest.validate() == 0
        ex = [x.cpu().numpy() for x in e.drain_examples()]
        meta = ex[5]
        order = np.lexsort((meta[:, 2], meta[:, 1], meta[:, 0]))
        res.append([x[order] for x in ex])
        e.close()
    assert len(res[0][0]) > 2 * T
    for a, b in zip(res[0], res[1]):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_selfplay_engine_default_mode_and_deterministic_switch():
    """the pipeline's default is the work-sharing budget; deterministic=True / AZG_DETERMINISTIC=1 selects the per-tree budget"""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetPipeline
    g = games.SplendorGame(2)
    args = Args(numMCTSSims=16, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.0, 1.0, 1.0], tempThreshold=6,
                **{**MCTS_ARGS['splendor2'], 'forced_playouts': False})
    for det in (None, True):
        e = SelfPlayEngine(g, HashNetPipeline(2), args, 32, node_capacity=1024, max_examples=32 * 400, deterministic=det)
        assert e.async_pipe and e.groups[0].async_cfg['shared_budget'] is (not det)
        e.start()
        e.run(200)
        st = e.stats()
        assert st['errors'] == 0 and st['plies'] > 0 and e.forest.validate() == 0
        e.close()


def test_async_pipeline_recovers_from_a_launch_that_ends_early(monkeypatch):
    """A launch of the pipeline that stalls (on this hardware: the platform freezes a few workgroups about once per 45 s of pipeline time,
    csrc/azg_async.hip.h "Recovery") ends early instead of hanging: a wave that idles past the time-out raises the abort flag, everyone leaves,
    and the NEXT launch re-queues the leaves that were never evaluated.  AZG_ASYNC_TEST_STALL cuts the 3rd launch short (time-out 20 us):
    the games played are still exactly the per-tree-budget run's, no error flag, and the early end is counted."""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetPipeline
    g = games.SplendorGame(2)
    T, sims = 48, 24
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
                **{**MCTS_ARGS['splendor2'], 'forced_playouts': False})
    res, early = [], 0
    for shared in (True, False):
        if shared:
            monkeypatch.setenv('AZG_ASYNC_TEST_STALL', '3')
        else:
            monkeypatch.delenv('AZG_ASYNC_TEST_STALL', raising=False)
        e = SelfPlayEngine(g, HashNetPipeline(2), args, T, node_capacity=2048, max_examples=T * 800, rng_seed=77, stream0=900,
                           async_pipe=True, async_cfg=dict(shared_budget=shared))
        e.start(episode_quota=2 * T)
        for k in range(4000):
            e.run(97)
            st = e.stats()
            assert st['errors'] == 0, st
            if st['active'] == 0:
                break
        assert st['games'] == 2 * T and e.forest.validate() == 0
        if shared:
            to = e.forest.async_profile(reset=False)['timeouts']
            early = to['select'] + to['net']
        ex = [x.cpu().numpy() for x in e.drain_examples()]
        meta = ex[5]
        order = np.lexsort((meta[:, 2], meta[:, 1], meta[:, 0]))
        res.append([x[order] for x in ex])
        e.close()
    assert early >= 1                                     # the third launch did end early ...
    for a, b in zip(res[0], res[1]):                      # ... and nothing was lost or played differently
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.gpu
def test_async_pipeline_per_tree_budget_survives_an_early_end(monkeypatch):
    """Per-tree budgets (SelfPlayEngine(deterministic=True)): a launch that ends early leaves every tree its unfinished calls (`carry`,
    csrc/azg_async.hip.h "Recovery") and the catch-up launch that follows every per-tree launch runs them -- so after each run() every tree
    has had exactly `rounds` calls again, stall or no stall.  AZG_ASYNC_TEST_STALL cuts the 3rd and the 7th launch of the process short
    (= the main launches of the 2nd and 4th run()): the engine's state after EVERY run() -- statistics counters, root statistics -- and the
    examples equal an undisturbed engine's."""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetPipeline
    g = games.SplendorGame(2)
    T, sims = 48, 24
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
                **{**MCTS_ARGS['splendor2'], 'forced_playouts': False})
    runs = []
    for stall in ('3', None):
        if stall:
            monkeypatch.setenv('AZG_ASYNC_TEST_STALL', stall)
        else:
            monkeypatch.delenv('AZG_ASYNC_TEST_STALL', raising=False)
        e = SelfPlayEngine(g, HashNetPipeline(2), args, T, node_capacity=2048, max_examples=T * 800, rng_seed=78, stream0=1200,
                           async_pipe=True, deterministic=True)
        e.start(episode_quota=T)
        trace = []
        for k in range(40):
            e.run(61)
            st = e.stats()
            assert st['errors'] == 0, st
            trace.append((st['games'], st['plies'], st['sims'], st['active']) + tuple(int(x) for x in e.forest.root_stats()['Ns'].cpu().numpy()))
            if k == 1 and stall:
                monkeypatch.setenv('AZG_ASYNC_TEST_STALL', '0')          # (the launch counter restarts: no further stall)
        to = e.forest.async_profile(reset=False)['timeouts']
        ex = [x.cpu().numpy() for x in e.drain_examples()]
        meta = ex[5]
        order = np.lexsort((meta[:, 2], meta[:, 1], meta[:, 0]))
        runs.append((trace, [x[order] for x in ex], to['select'] + to['net']))
        assert e.forest.validate() == 0
        e.close()
    assert runs[0][2] >= 1 and runs[1][2] == 0                     # the stalled engine did end a launch early
    assert runs[0][0] == runs[1][0]                                # ... and was where the undisturbed one was after every run()
    for a, b in zip(runs[0][1], runs[1][1]):
        assert a.shape == b.shape and np.array_equal(a, b)
