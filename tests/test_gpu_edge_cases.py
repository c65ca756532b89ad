"""GPU edge cases through the C-ABI: empty / ragged batches, error flags (arena, row-heap overflow), invalid arguments,
Splendor-specific corner states (int8 deck bitfield 255 -> -1, move counter > 127 with 3-4 players, deck exhaustion)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Args(dict):
    __getattr__ = dict.get


def test_empty_and_ragged_env_batches():
    import torch
    from azg_amd import games
    g = games.SplendorGame(2)
    dev = g.device
    empty = torch.zeros((0, g.S), dtype=torch.int8, device=dev)
    pl = torch.zeros((0,), dtype=torch.int32, device=dev)
    assert g.valid_moves_batch(empty, pl).shape == (0, 81)
    assert g.canonical_batch(empty, pl).shape == (0, g.S)
    for n in (1, 3, 65, 1000):                     # not multiples of anything
        b = g.init_boards_batch(n, stream0=7)
        v = g.valid_moves_batch(b, torch.zeros(n, dtype=torch.int32, device=dev))
        assert v.shape == (n, 81) and bool((v[:, 80] == 1).all())
        assert torch.equal(b[:1], g.init_boards_batch(1, stream0=7))          # stream i is independent of the batch size


def test_invalid_arguments_raise():
    from azg_amd import _lib, games
    from azg_amd.forest import Forest
    with pytest.raises(_lib.AzgError):
        _lib.game_info(_lib.SPLENDOR, 7)
    with pytest.raises(_lib.AzgError):
        _lib.game_info(99, 0)
    g = games.SplendorGame(2)
    with pytest.raises(_lib.AzgError):
        Forest(g.GAME_ID, g.variant, 4, Args(numMCTSSims=10, universes=99), node_capacity=64)
    with pytest.raises(_lib.AzgError):
        Forest(g.GAME_ID, g.variant, 0, Args(numMCTSSims=10), node_capacity=64)


def test_arena_overflow_is_reported_not_silent():
    """node_capacity smaller than numMCTSSims: the forest must flag the overflow (no corruption, no fallback)."""
    import torch
    from azg_amd import _lib, games
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    m = BatchedMCTS(g, HashNetTorch(2), Args(numMCTSSims=200, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True),
                    2, node_capacity=64)
    roots = g.init_boards_batch(2, stream0=1)
    with pytest.raises(_lib.AzgError, match='overflow'):
        m.getActionProb(roots, temp=1, force_full_search=True)
    m.forest.close()


@pytest.mark.parametrize('players', [3, 4])
def test_splendor_corner_states_vs_oracle(players):
    """Long random games on the GPU env vs the oracle: reaches the move cap (counter wraps past 127 in the int8 state),
    full decks stored as bitfield 255 -> -1, and tier-1 deck exhaustion."""
    import torch
    import azg_oracle as O
    from azg_amd import games
    g = games.SplendorGame(players)
    og = O.OracleGame(O.SPLENDOR, players)
    rng = np.random.default_rng(players)
    n = 64
    boards = np.stack([og.getInitBoard(og.rng(seed=5, stream=i)).reshape(-1) for i in range(n)])
    player = np.zeros(n, dtype=np.int32)
    seen_neg_counter = seen_minus1_bitfield = False
    for ply in range(62 * players + 4):
        b_dev = torch.from_numpy(boards).to(g.device)
        p_dev = torch.from_numpy(player).to(g.device)
        valid = g.valid_moves_batch(b_dev, p_dev).cpu().numpy()
        acts = np.zeros(n, dtype=np.int32)
        for i in range(n):
            v = og.getValidMoves(boards[i], int(player[i]))
            assert np.array_equal(valid[i].astype(bool), v), (players, ply, i)
            idx = np.flatnonzero(v)
            buy = idx[idx < 27]
            acts[i] = int(rng.choice(buy)) if len(buy) and rng.random() < 0.7 else int(rng.choice(idx))
        seeds = torch.full((n,), 31416 + ply, dtype=torch.int64, device=g.device)
        nb, npl = g.next_state_batch(b_dev, p_dev, torch.from_numpy(acts).to(g.device), seeds)
        nb, npl = nb.cpu().numpy(), npl.cpu().numpy()
        ended, scores, rnd = g.game_ended_batch(torch.from_numpy(nb).to(g.device), torch.from_numpy(npl).to(g.device))
        ended = ended.cpu().numpy()
        for i in range(n):
            eb, ep = og.getNextState(boards[i], int(player[i]), int(acts[i]), random_seed=31416 + ply)
            assert np.array_equal(nb[i], eb.reshape(-1)) and npl[i] == ep, (players, ply, i)
            assert np.array_equal(ended[i], og.getGameEnded(eb, ep))
        seen_neg_counter |= bool((nb[:, 6] < 0).any())
        seen_minus1_bitfield |= bool((nb.reshape(n, -1, 7)[:, 26, :5] == -1).any())
        boards, player = nb, npl.astype(np.int32)
    assert seen_neg_counter, 'move counter never exceeded 127'
    assert seen_minus1_bitfield, 'full tier-1 deck bitfield (255 -> -1) never seen'
