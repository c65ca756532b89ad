"""GPU parity: BatchedArena (Arena.playGames on the engine) vs the same match played move by move with the oracle
(two OracleMCTS contestants, the RNG contract's stream per game for init_game and the random_seed == 0 moves)."""
import numpy as np
import pytest

from tools_args import MCTS_ARGS

pytestmark = pytest.mark.gpu


class Args(dict):
    __getattr__ = dict.get


def test_batched_arena_vs_oracle():
    import torch
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.arena import BatchedArena
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    og = O.OracleGame(O.SPLENDOR, 2)
    kw = dict(MCTS_ARGS['splendor2'])
    sims1, sims2, T, stream0 = 24, 12, 8, 300
    a1 = Args(numMCTSSims=sims1, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    a2 = Args(numMCTSSims=sims2, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    arena = BatchedArena(g, HashNetTorch(2), HashNetTorch(2), a1, a2, n_parallel=T, node_capacity=2048, stream0=stream0)
    rec = []
    res, ovt = arena.play_wave(0, T, record=rec)
    res, ovt = res.cpu().numpy(), ovt.cpu().numpy()
    gpu_actions = [[] for _ in range(T)]
    for boards, cur, actions, done in rec:
        d, a = done.cpu().numpy(), actions.cpu().numpy()
        for i in range(T):
            if not d[i]:
                gpu_actions[i].append(int(a[i]))
    for i in range(T):
        rng = og.rng(seed=0, stream=stream0 + i)
        board, cur = og.getInitBoard(rng), 0
        ms = [O.OracleMCTS(og, O.make_args(numMCTSSims=sims1, **kw)), O.OracleMCTS(og, O.make_args(numMCTSSims=sims2, **kw))]
        one_vs_two = i % 4 in (0, 3)
        assert bool(ovt[i]) == one_vs_two
        acts = []
        while not og.getGameEnded(board, cur).any():
            owner = 0 if ((cur == 0) == one_vs_two) else 1
            canon = og.getCanonicalForm(board, cur)
            p, _, _ = ms[owner].getActionProb(canon, temp=1, force_full_search=True)
            a = int(np.argmax(p))
            acts.append(a)
            board, cur = og.getNextState(board, cur, a, random_seed=0, rng=rng)
        assert acts == gpu_actions[i], (i, acts[:10], gpu_actions[i][:10])
        assert float(og.getGameEnded(board, cur)[0]) == float(res[i])
    one, two, draws = arena.playGames(T)
    exp_one = sum(1 for i in range(T) if res[i] == (1.0 if ovt[i] else -1.0))
    exp_two = sum(1 for i in range(T) if res[i] == (-1.0 if ovt[i] else 1.0))
    assert (one, two, draws) == (exp_one, exp_two, T - exp_one - exp_two)


def test_arena_shards_reproduce_the_whole_match():
    """a match dealt out by game index (Coach.learn over several ranks): games [2, 5) played by their own arena object with
    first_game_index = 2 end like games 2..4 of the arena that plays all of [0, 6)"""
    from azg_amd import games
    from azg_amd.arena import BatchedArena
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    kw = dict(MCTS_ARGS['splendor2'])
    a1 = Args(numMCTSSims=16, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    a2 = Args(numMCTSSims=8, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    temp = lambda n: 0.5 * 0.5 ** (n / 6.0)                      # noqa: E731  (reaches the random tie-break branch late in a game)
    whole = BatchedArena(g, HashNetTorch(2), HashNetTorch(2), a1, a2, n_parallel=6, node_capacity=1024, stream0=1 << 32, temp_for_game=temp)
    res_w, ovt_w = whole.play_wave(0, 6)
    part = BatchedArena(g, HashNetTorch(2), HashNetTorch(2), a1, a2, n_parallel=3, node_capacity=1024, stream0=1 << 32, temp_for_game=temp,
                        first_game_index=2)
    res_p, ovt_p = part.play_wave(2, 3)
    assert np.array_equal(res_w.cpu().numpy()[2:5], res_p.cpu().numpy()) and np.array_equal(ovt_w.cpu().numpy()[2:5], ovt_p.cpu().numpy())
    assert part.playGames(3, first_game_index=2) == tuple(
        int(x) for x in (((ovt_p & (res_p == 1.0)) | (~ovt_p & (res_p == -1.0))).sum(), ((ovt_p & (res_p == -1.0)) | (~ovt_p & (res_p == 1.0))).sum(),
                         (~((res_p == 1.0) | (res_p == -1.0))).sum()))
