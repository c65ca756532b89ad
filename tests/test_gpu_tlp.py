"""GPU parity of The Little Prince plugin: env kernels vs the reference's golden transitions (the engine draws the
same uniforms from its counter RNG that the reference consumed through tools/refshim's CounterRandom), the random symmetries on the
recorded streams, MCTS with search-time market refills vs the reference's golden traces, whole trees and self-play episodes vs the
pinned oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
KW = dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True)


class Args(dict):
    __getattr__ = dict.get


@pytest.mark.parametrize('n', [3, 4, 5])
def test_tlp_env_vs_golden(golden_dir, n):
    from azg_amd import games
    d = np.load(os.path.join(golden_dir, 'env_tlp%d.npz' % n))
    g = games.TLPGame(n, rng_seed=2000 + n)                          # the seed tools/gen_golden_tlp.py drew from
    dev = g.device
    assert (g.S, g.A, g.P) == (d['state'].shape[1], n * n, n) and tuple(g.getBoardSize()) == tuple(d['shape'])
    T = len(d['init_boards'])
    counters = torch.zeros(T, dtype=torch.int64, device=dev)
    boards = g.init_boards_batch(T, stream0=0, counters=counters)    # stream t = trajectory t, like the generator
    assert np.array_equal(boards.cpu().numpy(), d['init_boards']) and (counters == 1 + n).all()
    state, player = torch.from_numpy(d['state']).to(dev), torch.from_numpy(d['player'].astype(np.int32)).to(dev)
    valid = g.valid_moves_batch(state, player).cpu().numpy()
    assert np.array_equal(np.packbits(valid, axis=1), d['valid'])
    traj = d['traj']
    idx_of = [np.flatnonzero(traj == t) for t in range(T)]
    zero = torch.zeros(T, dtype=torch.int64, device=dev)
    for k in range(max(len(ix) for ix in idx_of)):
        rows = np.array([ix[k] if k < len(ix) else ix[-1] for ix in idx_of])
        live = np.array([k < len(ix) for ix in idx_of])
        before = counters.clone()
        nb, npl = g.next_state_batch(torch.from_numpy(d['state'][rows]).to(dev), torch.from_numpy(d['player'][rows].astype(np.int32)).to(dev),
                                     torch.from_numpy(d['action'][rows].astype(np.int32)).to(dev), zero, stream0=0, counters=counters)
        counters = torch.where(torch.from_numpy(live).to(dev), counters, before)
        nbn, npn, used = nb.cpu().numpy(), npl.cpu().numpy(), (counters - before).cpu().numpy()
        assert np.array_equal(nbn[live], d['next_state'][rows][live]), (n, k)
        assert np.array_equal(npn[live], d['next_player'][rows][live].astype(np.int32))
        assert np.array_equal(used[live], d['n_uniforms'][rows][live].astype(np.int64))     # same number of draws consumed
    ns, npl = torch.from_numpy(d['next_state']).to(dev), torch.from_numpy(d['next_player'].astype(np.int32)).to(dev)
    ended, score, rnd = g.game_ended_batch(ns, npl)
    assert np.array_equal(ended.cpu().numpy(), d['ended']) and np.array_equal(score.cpu().numpy(), d['score'].astype(np.int32))
    assert np.array_equal(rnd.cpu().numpy(), d['round'].astype(np.int32))
    assert np.array_equal(g.canonical_batch(ns, npl).cpu().numpy(), d['canonical'])
    b0 = d['state'][5].reshape(g.getBoardSize())
    assert np.array_equal(np.packbits(g.getValidMoves(b0, int(d['player'][5])).astype(np.uint8)), d['valid'][5])


@pytest.mark.parametrize('n', [3, 4, 5])
def test_tlp_random_symmetries_vs_golden(golden_dir, n):
    """get_symmetries (np.random.shuffle of players / market cards / planet slots, duplicate states dropped) on the streams the
    reference drew from: same forms in the same order"""
    from azg_amd import games
    d = np.load(os.path.join(golden_dir, 'sym_tlp%d.npz' % n))
    g = games.TLPGame(n)
    dev = g.device
    K = d['out_state'].shape[1]
    assert K == g.max_symmetries()
    ob, op, ov, cnt = g.symmetries_batch(torch.from_numpy(d['state']).to(dev), torch.from_numpy(d['pi']).to(dev),
                                         torch.from_numpy(d['valids']).to(dev), rng_seed=int(d['seed']), stream0=0)
    cnt = cnt.cpu().numpy()
    assert np.array_equal(cnt, d['count'])
    ob, op, ov = ob.cpu().numpy(), op.cpu().numpy(), ov.cpu().numpy()
    for j in range(len(cnt)):
        k = int(cnt[j])
        assert np.array_equal(ob[j, :k], d['out_state'][j, :k]), j
        assert np.array_equal(op[j, :k], d['out_pi'][j, :k]) and np.array_equal(ov[j, :k], d['out_valids'][j, :k])
    # Game.py surface: the identity comes first, forms are distinct
    b0 = d['state'][7].reshape(g.getBoardSize())
    syms = g.getSymmetries(b0, d['pi'][7], d['valids'][7].astype(bool))
    assert np.array_equal(syms[0][0], b0) and len({s.tobytes() for s, _, _ in syms}) == len(syms)


def test_tlp_mcts_vs_golden(golden_dir):
    from azg_amd import games
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, 'mcts_tlp3_numba.npz'))
    g = games.TLPGame(3)
    for i in range(len(d['case_sims'])):
        sims = int(d['case_sims'][i])
        args = Args(numMCTSSims=sims, cpuct=float(d['case_cpuct'][i]), fpu=float(d['case_fpu'][i]), universes=int(d['case_universes'][i]),
                    forced_playouts=bool(d['case_forced'][i]), prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1])
        m = BatchedMCTS(g, HashNetTorch(3), args, 1, node_capacity=sims + 64, rng_seed=int(d['case_rng_seed'][i]),
                        stream0=int(d['case_rng_stream'][i]))
        probs, q, _ = m.getActionProb(torch.from_numpy(d['case_root'][i:i + 1]).to(g.device), temp=1, force_full_search=True)
        rs = m.forest.root_stats()
        assert int(rs['Ns'][0]) == int(d['case_Ns'][i]) and int(rs['n_nodes'][0]) == int(d['case_nodes'][i]), i
        assert np.array_equal(rs['Nsa'][0].cpu().numpy(), d['case_Nsa'][i].astype(np.int32)), i
        assert np.array_equal(rs['Qsa'][0].cpu().numpy(), d['case_Qsa'][i]) and float(rs['Qs'][0]) == float(d['case_Qs'][i])
        assert np.array_equal(probs[0].cpu().numpy(), d['case_probs'][i]) and np.array_equal(q[0].cpu().numpy(), d['case_q'][i])
        m.forest.close()


@pytest.mark.parametrize('n', [3, 4, 5])
def test_tlp_whole_tree_and_selfplay_vs_oracle(n):
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    g, og = games.TLPGame(n), O.OracleGame(O.TLP, n)
    # ---- every node of the tree, 300 simulations, 8 trees on their own streams ----
    T, sims, seed, stream0 = 8, 300, 31, 500
    roots = np.stack([og.getInitBoard(og.rng(seed=9, stream=i)).reshape(-1) for i in range(T)])
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **KW)
    m = BatchedMCTS(g, HashNetTorch(n), args, T, node_capacity=sims + 64, rng_seed=seed, stream0=stream0)
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
    # ---- Coach.executeEpisode: search refills, move picks and real refills interleave on one stream per game ----
    sims, T, seed, stream0, temp = 30, 16, 4242, 1000, [1.25, 0.8, 1.0]
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=temp, tempThreshold=6, **KW)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=4096, max_examples=T * 400, rng_seed=seed, stream0=stream0)
    net = HashNetTorch(n)
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
        assert len(sel) == o['plies'] == 16 * n, (t, len(sel), o['plies'])
        for k, ply in zip(sel, range(o['plies'])):
            assert meta[k, 3] == o['player'][ply] and np.array_equal(boards[k], o['canonical'][ply]), (t, ply)
            assert np.array_equal(pis[k], o['pi'][ply].astype(np.float32)) and np.array_equal(qs[k], o['q'][ply])
            r = o['result']
            assert np.array_equal(zs[k], np.roll(r, -int(o['player'][ply])))
    f.close()


def test_tlp_engine_symmetric_drain():
    """SelfPlayEngine.drain_examples(symmetries=True): every record expands into its distinct random forms, identity first"""
    from azg_amd import games
    from azg_amd.selfplay import SelfPlayEngine
    from hashnet import HashNetTorch
    g = games.TLPGame(3, rng_seed=5)
    args = Args(numMCTSSims=20, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.25, 0.8, 1.0], tempThreshold=6, **KW)
    eng = SelfPlayEngine(g, HashNetTorch(3), args, n_games=32, node_capacity=2048, max_examples=32 * 100, use_graph=False)
    eng.start(episode_quota=32)
    for _ in range(200):
        eng.run(rounds=200)
        if eng.stats()['active'] == 0:
            break
    st = eng.stats()
    assert st['errors'] == 0 and st['games'] == 32, st
    plain = eng.drain_examples()
    assert plain[0].shape[0] == 32 * 48
    sym = eng.drain_examples(symmetries=True)
    assert sym[0].shape[0] == 0                                     # (already drained)
    ob, op, ov, cnt = g.symmetries_batch(plain[0].contiguous(), plain[1].contiguous(), plain[3].contiguous(), stream0=77)
    cnt = cnt.cpu().numpy()
    assert cnt.min() >= 1 and cnt.max() <= 7 and cnt.max() >= 3
    assert np.array_equal(ob[:, 0].cpu().numpy(), plain[0].cpu().numpy()) and np.array_equal(op[:, 0].cpu().numpy(), plain[1].cpu().numpy())
