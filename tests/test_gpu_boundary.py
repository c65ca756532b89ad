"""GPU: the reference-shaped plugin surface on the engine -- the single-tree `MCTS` class (MCTS.py:24,49), a net that only offers
the per-sample `predict` (NeuralNet.py:32-43), root noise and the low-temperature branch of getActionProb (MCTS.py:64,93-98), an
Arena.playGame-shaped loop (Arena.py:52-101) over the Game / MCTS mirrors, and the NNetWrapper / Coach call sequences of
main.py / pit.py."""
import os

import numpy as np
import pytest
import torch

from tools_args import MCTS_ARGS

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


class Args(dict):
    __getattr__ = dict.get


def test_single_tree_mcts_class_with_per_sample_predict_vs_golden():
    """MCTS(game, nnet, args).getActionProb(board, temp, force_full_search) with a net that has only predict(): the returned
    (probs list, q list, is_full) equal the reference's own outputs (mcts_splendor2_numba.npz, 25-simulation cases)."""
    from azg_amd import games
    from azg_amd.mcts import MCTS
    from hashnet import HashNetNumpy
    d = np.load(os.path.join(GOLDEN, 'mcts_splendor2_numba.npz'))
    g = games.SplendorGame(2)
    done = 0
    for i in range(len(d['case_sims'])):
        if int(d['case_sims'][i]) != 25:
            continue
        args = Args(numMCTSSims=25, cpuct=float(d['case_cpuct'][i]), fpu=float(d['case_fpu'][i]), universes=int(d['case_universes'][i]),
                    forced_playouts=bool(d['case_forced'][i]), prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3,
                    temperature=[1, 1, 1], no_mem_optim=False)
        net = HashNetNumpy(2)
        m = MCTS(g, net, args)                                         # dirichlet_noise defaults to False: no noise despite alpha
        probs, q, full = m.getActionProb(d['case_root'][i].reshape(g.getBoardSize()), temp=1, force_full_search=True)
        assert isinstance(probs, list) and len(probs) == g.getActionSize() and full is True
        assert np.array_equal(np.asarray(probs), d['case_probs'][i]) and np.array_equal(np.asarray(q, dtype=np.float32), d['case_q'][i])
        assert net.calls == int(d['case_nodes'][i])                    # one predict() per expanded node (MCTS.py:144)
        m._b.forest.close()
        done += 1
    assert done >= 4


def test_mcts_class_root_noise_and_low_temperature():
    """dirichlet_noise=True (Coach.py:31,96): the root prior of a full search becomes 0.75 P + 0.25 Dir (MCTS.py:64,187-197) --
    a proper distribution over the valid actions that differs from the noise-free prior; temp <= 0.02 (MCTS.py:93-98): a one-hot
    on a most-visited action, drawn among ties with the tree's RNG stream."""
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    og = O.OracleGame(O.SPLENDOR, 2)
    T = 64
    roots = torch.from_numpy(np.stack([og.getInitBoard(og.rng(seed=3, stream=i)).reshape(-1) for i in range(T)])).to(g.device)
    args = Args(numMCTSSims=30, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
                **MCTS_ARGS['splendor2'])
    plain = BatchedMCTS(g, HashNetTorch(2), args, T)
    noisy = BatchedMCTS(g, HashNetTorch(2), args, T, dirichlet_noise=True)
    plain.getActionProb(roots, temp=1, force_full_search=True)
    noisy.getActionProb(roots, temp=1, force_full_search=True)
    p0, p1 = plain.forest.root_stats()['Ps'].cpu().numpy(), noisy.forest.root_stats()['Ps'].cpu().numpy()
    assert np.allclose(p1.sum(axis=1), 1.0, atol=1e-5) and np.all((p1 > 0) == (p0 > 0))
    assert np.all(np.abs(p1 - p0).max(axis=1) > 1e-3)                  # every root got its own noise
    assert np.all(p1[p0 > 0] >= 0.75 * p0[p0 > 0] / 1.01 - 1e-6)        # 0.75 P + 0.25 Dir, renormalised (sum ~ 1)
    # fast searches get no noise (MCTS.py:64 `is_full_search`)
    fast = BatchedMCTS(g, HashNetTorch(2), args, T, dirichlet_noise=True)
    fast.getActionProb(roots, temp=1, full=torch.zeros(T, dtype=torch.uint8, device=g.device))
    plain_fast = BatchedMCTS(g, HashNetTorch(2), args, T)
    plain_fast.getActionProb(roots, temp=1, full=torch.zeros(T, dtype=torch.uint8, device=g.device))
    assert np.array_equal(fast.forest.root_stats()['Ps'].cpu().numpy(), plain_fast.forest.root_stats()['Ps'].cpu().numpy())
    # ---- temp <= 0.02 ----
    probs1, _, _ = plain.forest.action_probs(1.0)
    probs0, _, _ = plain.forest.action_probs(0.01)
    probs0b, _, _ = plain.forest.action_probs(0.0)
    p1n, p0n = probs1.cpu().numpy(), probs0.cpu().numpy()
    ties, picks_not_first = 0, 0
    for t in range(T):
        best = np.flatnonzero(p1n[t] == p1n[t].max())
        assert p0n[t].sum() == 1.0 and p0n[t].max() == 1.0 and int(np.argmax(p0n[t])) in best
        if len(best) > 1:
            ties += 1
            picks_not_first += int(np.argmax(p0n[t]) != best[0])
    # 30 simulations over ~40 valid actions leave ties between the most visited actions in some trees; the pick must not always be
    # the first one (np.random.choice over the maxima in the reference)
    if ties >= 8:
        assert picks_not_first > 0
    assert np.array_equal(probs0b.cpu().numpy().sum(axis=1), np.ones(T))
    for m in (plain, noisy, fast, plain_fast):
        m.forest.close()


def test_arena_playgame_loop_over_the_mirrors_vs_oracle():
    """Arena.playGame (Arena.py:52-101) restated over azg_amd.games.SantoriniGame + azg_amd.mcts.MCTS players built like
    pit.py:58-64 (np.argmax of getActionProb(x, temp=temp_for_game(n), force_full_search=True)), against the same loop driven
    by the oracle's MCTS: same actions at every turn, same result."""
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.mcts import MCTS
    from hashnet import HashNetNumpy
    g = games.SantoriniGame(1)
    og = O.OracleGame(O.SANTORINI, 1)
    kw1 = dict(MCTS_ARGS['santorini1'])
    kw2 = dict(kw1, cpuct=1.6, fpu=0.1)
    sims = 24

    def temp_for_game(n):                                              # pit.py:58-61 with half_life 10
        return 0.5 * (0.5 ** (n / 10))
    mk = lambda kw: Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, no_mem_optim=False, **kw)     # noqa: E731
    m1, m2 = MCTS(g, HashNetNumpy(2), mk(kw1)), MCTS(g, HashNetNumpy(2), mk(kw2))
    o1, o2 = O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **kw1)), O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **kw2))
    players = [lambda x, n: int(np.argmax(m1.getActionProb(x, temp=temp_for_game(n), force_full_search=True)[0])),
               lambda x, n: int(np.argmax(m2.getActionProb(x, temp=temp_for_game(n), force_full_search=True)[0]))]
    oracles = [o1, o2]
    board = g.getInitBoard()
    assert np.array_equal(board.reshape(-1), og.getInitBoard(og.rng(seed=0, stream=1)).reshape(-1))
    oboard = og.getInitBoard(og.rng(seed=0, stream=1))
    cur, it = 0, 0
    while not g.getGameEnded(board, cur).any():
        it += 1
        canonical = g.getCanonicalForm(board, cur)
        action = players[cur](canonical, it)
        valids = g.getValidMoves(canonical, 0)
        assert valids[action]
        temp = temp_for_game(it)
        oprobs, _, _ = oracles[cur].getActionProb(og.getCanonicalForm(oboard, cur), temp=max(temp, 0.05),
                                                   force_full_search=True)
        if temp > 0.02:
            assert action == int(np.argmax(oprobs)), (it, action)
        else:                                                          # random pick among the most visited (MCTS.py:93-98)
            assert oprobs[action] == oprobs.max(), (it, action)
        board, cur_next = g.getNextState(board, cur, action, random_seed=0)
        oboard, ocur = og.getNextState(oboard, cur, action, random_seed=0, rng=og.rng(seed=0, stream=0))
        assert np.array_equal(board.reshape(-1), oboard.reshape(-1)) and int(cur_next) == int(ocur)
        cur = int(cur_next)
        assert it < 300
    assert np.array_equal(g.getGameEnded(board, cur), og.getGameEnded(oboard, cur))
    assert it > 8
    MCTS.reset_all_search_trees()                                      # Arena.py:99
    assert int(m1._b.forest.root_stats()['n_nodes'][0]) == 0


def test_nnet_wrapper_predict_checkpoints_and_coach_call_sequence(tmp_path):
    """main.py / pit.py call sequences: NNetWrapper(game, nn_args) -> load weights -> predict(board, valids) equals the reference
    model's outputs; save_checkpoint(folder, file, additional_keys) / load_checkpoint -> dict; Coach(game, nnet, args).learn()."""
    from azg_amd import games
    from azg_amd.coach import Coach
    from azg_amd.nnet_wrapper import NNetWrapper
    g = games.SplendorGame(2)
    nn_args = dict(lr=1e-3, learn_rate=1e-3, dropout=0., epochs=1, batch_size=64, nn_version=80, q_weight=0.5)
    w = NNetWrapper(g, nn_args)
    z = np.load(os.path.join(GOLDEN, 'weights_splendor2_v80.npz'))
    w.nnet.load_state_dict({k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd/')}, strict=True)
    d = np.load(os.path.join(GOLDEN, 'netfwd_splendor2_v80.npz'))
    for i in range(0, 16):
        pi, v = w.predict(d['boards'][i], d['masks'][i].astype(bool))
        assert pi.shape == (81,) and v.shape == (2,) and pi.dtype == np.float32
        assert np.abs(pi - d['pi'][i]).max() <= 1e-5 and np.abs(v - d['v'][i]).max() <= 1e-5
    w.save_checkpoint(str(tmp_path), 'a.pt', additional_keys=dict(cpuct=0.8, numMCTSSims=25, temperature=[1.25, 0.8]))
    p = NNetWrapper(g, dict(lr=None, dropout=0., epochs=None, batch_size=None, nn_version=-1))                # pit.py:44
    keys = p.load_checkpoint(str(tmp_path), 'a.pt')
    assert keys['cpuct'] == 0.8 and keys['full_model'].version == 80
    pi2, v2 = p.predict(d['boards'][3], d['masks'][3].astype(bool))
    pi1, v1 = w.predict(d['boards'][3], d['masks'][3].astype(bool))
    assert np.array_equal(pi1, pi2) and np.array_equal(v1, v2)
    assert p.load_checkpoint(str(tmp_path), 'missing.pt') is None
    args = Args(numMCTSSims=8, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=False, dirichletAlpha=0.3, prob_fullMCTS=1.0,
                ratio_fullMCTS=5, temperature=[1.25, 0.8, 1.0], tempThreshold=6, numIters=1, numEps=12, numItersHistory=2,
                maxlenOfQueue=100000, arenaCompare=4, updateThreshold=0.6, checkpoint=str(tmp_path), n_games=8,
                stop_after_N_fail=5, load_folder_file=[str(tmp_path), 'best.pt'])
    c = Coach(g, w, args)                                              # main.py:60-66
    c.log = lambda s: None
    c.cap = 1024
    res = c.learn()
    assert len(res) == 1 and res[0]['nwins'] + res[0]['pwins'] + res[0]['draws'] == 4
    assert os.path.exists(os.path.join(tmp_path, 'temp.pt')) and os.path.exists(os.path.join(tmp_path, 'checkpoint.examples'))
    c2 = Coach(g, NNetWrapper(g, nn_args), args)
    c2.loadTrainExamples()                                             # main.py:64-65
    assert len(c2.trainExamplesHistory) == 1 and len(c2.trainExamplesHistory[0]) == len(c.trainExamplesHistory[0])
    assert c.getCheckpointFile(3) == 'checkpoint_3.pt' and abs(c.temp_for_game(6) - 0.25) < 1e-12
