"""GPU parity of the Minivilles plugin: env kernels vs the reference's golden transitions (the engine draws the
same uniforms from its counter RNG that the reference consumed through tools/refshim's CounterRandom), MCTS with search-time dice vs
the reference's golden traces, whole trees and self-play episodes vs the pinned oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True)


class Args(dict):
    __getattr__ = dict.get


@pytest.mark.parametrize('n', [2, 3, 4])
def test_minivilles_env_vs_golden(golden_dir, n):
    from azg_amd import games
    d = np.load(os.path.join(golden_dir, 'env_minivilles%d.npz' % n))
    g = games.MinivillesGame(n, rng_seed=1000 + n)                   # the seed tools/gen_golden_minivilles.py drew from
    dev = g.device
    assert (g.S, g.A, g.P) == (d['state'].shape[1], 21, n) and tuple(g.getBoardSize()) == tuple(d['shape'])
    T = len(d['init_boards'])
    counters = torch.zeros(T, dtype=torch.int64, device=dev)
    boards = g.init_boards_batch(T, stream0=0, counters=counters)    # stream t = trajectory t, like the generator
    assert np.array_equal(boards.cpu().numpy(), d['init_boards'])
    state, player = torch.from_numpy(d['state']).to(dev), torch.from_numpy(d['player'].astype(np.int32)).to(dev)
    valid = g.valid_moves_batch(state, player).cpu().numpy()
    assert np.array_equal(np.packbits(valid, axis=1), d['valid'])
    # transitions: step k of every trajectory in one launch (item t draws from stream t at its running counter)
    traj = d['traj']
    idx_of = [np.flatnonzero(traj == t) for t in range(T)]
    zero = torch.zeros(T, dtype=torch.int64, device=dev)
    for k in range(max(len(ix) for ix in idx_of)):
        rows = np.array([ix[k] if k < len(ix) else ix[-1] for ix in idx_of])
        live = np.array([k < len(ix) for ix in idx_of])
        before = counters.clone()
        nb, npl = g.next_state_batch(torch.from_numpy(d['state'][rows]).to(dev), torch.from_numpy(d['player'][rows].astype(np.int32)).to(dev),
                                     torch.from_numpy(d['action'][rows].astype(np.int32)).to(dev), zero, stream0=0, counters=counters)
        counters = torch.where(torch.from_numpy(live).to(dev), counters, before)          # finished trajectories keep their counter
        nbn, npn, used = nb.cpu().numpy(), npl.cpu().numpy(), (counters - before).cpu().numpy()
        assert np.array_equal(nbn[live], d['next_state'][rows][live]), (n, k)
        assert np.array_equal(npn[live], d['next_player'][rows][live].astype(np.int32))
        assert np.array_equal(used[live], d['n_uniforms'][rows][live].astype(np.int64))     # same number of draws consumed
    ns, npl = torch.from_numpy(d['next_state']).to(dev), torch.from_numpy(d['next_player'].astype(np.int32)).to(dev)
    ended, score, rnd = g.game_ended_batch(ns, npl)
    assert np.array_equal(ended.cpu().numpy(), d['ended']) and np.array_equal(score.cpu().numpy(), d['score'].astype(np.int32))
    assert np.array_equal(rnd.cpu().numpy(), d['round'].astype(np.int32))
    assert np.array_equal(g.canonical_batch(ns, npl).cpu().numpy(), d['canonical'])
    # Game.py surface, one board
    b0 = d['state'][5].reshape(g.getBoardSize())
    assert np.array_equal(np.packbits(g.getValidMoves(b0, int(d['player'][5])).astype(np.uint8)), d['valid'][5])
    syms = g.getSymmetries(b0, np.full(21, 1 / 21, dtype=np.float32), np.ones(21, dtype=bool))
    assert len(syms) == 1 and np.array_equal(syms[0][0], b0)


def test_minivilles_mcts_vs_golden(golden_dir):
    """MCTS.getActionProb with the dice of every simulated step drawn inside the search (no edge is memoised): root statistics,
    probs, q and node count equal the reference's, case by case, on the stream the reference drew from."""
    from azg_amd import games
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, 'mcts_minivilles2_numba.npz'))
    g = games.MinivillesGame(2)
    for i in range(len(d['case_sims'])):
        sims = int(d['case_sims'][i])
        args = Args(numMCTSSims=sims, cpuct=float(d['case_cpuct'][i]), fpu=float(d['case_fpu'][i]), universes=int(d['case_universes'][i]),
                    forced_playouts=bool(d['case_forced'][i]), prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1])
        m = BatchedMCTS(g, HashNetTorch(2), args, 1, node_capacity=sims + 64, rng_seed=int(d['case_rng_seed'][i]),
                        stream0=int(d['case_rng_stream'][i]))
        probs, q, _ = m.getActionProb(torch.from_numpy(d['case_root'][i:i + 1]).to(g.device), temp=1, force_full_search=True)
        rs = m.forest.root_stats()
        assert int(rs['Ns'][0]) == int(d['case_Ns'][i]) and int(rs['n_nodes'][0]) == int(d['case_nodes'][i]), i
        assert np.array_equal(rs['Nsa'][0].cpu().numpy(), d['case_Nsa'][i].astype(np.int32)), i
        assert np.array_equal(rs['Qsa'][0].cpu().numpy(), d['case_Qsa'][i]) and float(rs['Qs'][0]) == float(d['case_Qs'][i])
        assert np.array_equal(probs[0].cpu().numpy(), d['case_probs'][i]) and np.array_equal(q[0].cpu().numpy(), d['case_q'][i])
        m.forest.close()


def test_minivilles_whole_tree_and_selfplay_vs_oracle():
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    g, og = games.MinivillesGame(2), O.OracleGame(O.MINIVILLES, 2)
    # ---- every node of the tree, 300 simulations, 8 trees on their own streams ----
    T, sims, seed, stream0 = 8, 300, 31, 500
    roots = np.stack([og.getInitBoard(og.rng(seed=9, stream=i)).reshape(-1) for i in range(T)])
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **KW)
    m = BatchedMCTS(g, HashNetTorch(2), args, T, node_capacity=sims + 64, rng_seed=seed, stream0=stream0)
    m.getActionProb(torch.from_numpy(roots).to(g.device), temp=1, force_full_search=True)
    for t in range(T):
        om = O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **KW))
        om.set_rng(og.rng(seed=seed, stream=stream0 + t))
        om.getActionProb(roots[t], temp=1, force_full_search=True)
        tree = m.forest.dump_tree(t)
        assert tree['n'] == om.num_nodes()
        for i in range(tree['n']):
            nd = om.node(tree['states'][i])
            assert nd is not None and np.array_equal(nd['Es'], tree['Es'][i]) and nd['has_policy'] == bool(tree['has_policy'][i])
            if nd['has_policy']:
                assert nd['Ns'] == int(tree['Ns'][i]) and nd['Qs'] == tree['Qs'][i]
                assert np.array_equal(nd['Nsa'].astype(np.int32), tree['Nsa'][i]) and np.array_equal(nd['Qsa'], tree['Qsa'][i])
    m.forest.close()
    # ---- Coach.executeEpisode: search dice, move picks and real dice interleave on one stream per game ----
    sims, T, seed, stream0, temp = 30, 16, 4242, 1000, [1.25, 0.8, 1.0]
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=temp, tempThreshold=6, **KW)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=4096, max_examples=T * 700, rng_seed=seed, stream0=stream0)
    net = HashNetTorch(2)
    f.selfplay_start()
    for rnd in range(400000):
        f.select()
        pi, vv = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
        f.expand_backup(pi, vv)
        f.selfplay_advance()
        if rnd % 256 == 255:
            st = f.stats()
            assert st['errors'] == 0, st
            if st['games'] >= 2 * T:
                break
    assert f.validate() == 0
    boards, pis, zs, valids, qs, meta = [x.cpu().numpy() for x in f.drain_examples()]
    for t in range(T):
        o = O.run_episode(og, O.make_args(numMCTSSims=sims, **KW), None, seed=seed, stream=stream0 + t, temp=(temp[0], temp[1]),
                          tempThreshold=6.0)
        sel = np.flatnonzero((meta[:, 0] == stream0 + t) & (meta[:, 1] == 0))
        sel = sel[np.argsort(meta[sel, 2])]
        assert len(sel) == o['plies'], (t, len(sel), o['plies'])
        for k, ply in zip(sel, range(o['plies'])):
            assert meta[k, 3] == o['player'][ply] and np.array_equal(boards[k], o['canonical'][ply]), (t, ply)
            assert np.array_equal(pis[k], o['pi'][ply].astype(np.float32)) and np.array_equal(qs[k], o['q'][ply])
            r = o['result']
            assert np.array_equal(zs[k], np.roll(r, -int(o['player'][ply])))
    f.close()


@pytest.mark.parametrize('game_tag', [('minivilles', 'minivilles2_v82'), ('tlp', 'tlp3_v83')], ids=['minivilles_v82', 'tlp_v83'])
def test_selfplay_with_the_shipped_engine_net(golden_dir, game_tag):
    """Self-play of the two f4 games whose shipped nets are engine kernels (MobileNet1dHip on the reference's own checkpoints, the MCTS
    arguments stored with them): the engine's forest with the one-launch net and with the same weights as PyTorch ops plays the same
    first plies (the two evaluators agree to <= 1e-5, so a search may differ later where two PUCT scores tie within that), no engine
    errors, structurally valid trees, examples for every finished game."""
    from azg_amd import games, nnet
    from azg_amd.selfplay import SelfPlayEngine
    game, tag = game_tag
    z = np.load(os.path.join(golden_dir, 'weights_%s.npz' % tag))
    a = Args(numMCTSSims=50, cpuct=float(z['arg/cpuct']), fpu=float(z['arg/fpu']), universes=int(z['arg/universes']), forced_playouts=True,
             prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.0, temperature=[1.25, 0.8, 1.0], tempThreshold=4)
    T = 64
    out = []
    for engine_net in (True, False):
        g = games.MinivillesGame(2) if game == 'minivilles' else games.TLPGame(3)
        base = nnet.MobileNet1d.from_npz(os.path.join(golden_dir, 'weights_%s.npz' % tag), device='cuda:0')
        assert (base.nb_vect * base.L, base.A, base.P) == (g.S, g.A, g.P)
        net = nnet.MobileNet1dHip(base, max_batch=T) if engine_net else base
        eng = SelfPlayEngine(g, net, a, n_games=T, node_capacity=2048, max_examples=T * 256, use_graph=False)
        eng.start()
        eng.run(6 * 50)
        torch.cuda.synchronize()
        st = eng.stats()
        assert st['errors'] == 0 and st['plies'] >= 3 * T
        assert sum(grp.f.validate() for grp in eng.groups) == 0
        boards, pi, zz, valids, q, meta = eng.drain_examples(symmetries=False)
        pi, valids = torch.as_tensor(pi).cpu().numpy(), torch.as_tensor(valids).cpu().numpy()
        assert np.all(np.isfinite(pi)) and np.all(pi[valids == 0] == 0) and np.allclose(pi.sum(axis=1), 1.0, atol=1e-5)
        out.append((st['plies'], st['sims']))
        for grp in eng.groups:
            grp.f.close()
    assert abs(out[0][0] - out[1][0]) <= T        # the same pace of play with either evaluator
