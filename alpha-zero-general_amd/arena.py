"""Arena.playGames on the engine (Arena.py:35-140, pit.py:26-66): N head-to-head games run concurrently, one search tree
per game and per contestant, every ply one batched MCTS per contestant over the games where it is to move.

The reference's own Arena only needs the duck-typed surface, so `Arena.Arena(player1, player2, azg_amd.games.SplendorGame())`
with `azg_amd.mcts.MCTS`-based players already works unchanged (one game at a time); this class is the batched form:
  * seats alternate 1-2-2-1 over the games (Arena.py:121-125): game i is "one vs two" when i % 4 in (0, 3); with more than
    two players the first seat belongs to one contestant and all other seats to the other (Arena.py:52-55);
  * a contestant = (nnet, args): its move is argmax_a of getActionProb(canonical, temp=temp_for_game(turn),
    force_full_search=True) (Coach.py:193-194, pit.py:60-64): above 0.02 the temperature only sharpens the visit counts and the
    argmax is the most visited action (first index on ties, np.argmax); once temp_for_game(turn) <= 0.02 (late in long games)
    getActionProb itself returns a one-hot on a maximum drawn at random among ties (MCTS.py:93-98; the tree's counter RNG);
  * the real move uses random_seed = 0 (Arena.py:84) -- the engine's counter-based RNG stream of that game;
  * the result of a game is getGameEnded(board, curPlayer)[0] (Arena.py:101), tallied like playGames (:126-131)."""
import torch

from .mcts import BatchedMCTS


class BatchedArena:
    def __init__(self, game, nnet1, nnet2, args1, args2=None, n_parallel=64, node_capacity=None, stream0=0, temp_for_game=None,
                 first_game_index=0):
        """temp_for_game(turn) -> temperature of getActionProb at that turn (Coach.temp_for_game, Coach.py:273-276, or pit.py's
        variant); None = 1 at every turn (plain argmax of the visit counts).  first_game_index: index of the game tree 0 plays when
        a match is dealt out over several ranks -- the trees' own random streams (ties at temperature <= 0.02) follow the game index"""
        self.game, self.T, self.stream0 = game, n_parallel, stream0
        self.temp_for_game = temp_for_game
        kw = [dict(rng_seed=int(getattr(game, 'rng_seed', 0)), stream0=stream0 + (c + 1) * (1 << 30) + first_game_index) for c in (0, 1)]
        self.mcts = [BatchedMCTS(game, nnet1, args1, n_parallel, node_capacity=node_capacity, **kw[0]),
                     BatchedMCTS(game, nnet2, args2 if args2 is not None else args1, n_parallel, node_capacity=node_capacity, **kw[1])]
        self.max_plies = 4096

    def play_wave(self, first_game_index=0, n_games=None, record=None):
        """plays games [first_game_index, first_game_index + n_games) concurrently (n_games <= n_parallel);
        returns results f32[n_games] = getGameEnded(...)[0] per game and the bool 'one_vs_two' seating per game"""
        g, T, dev = self.game, self.T, self.game.device
        n = T if n_games is None else n_games
        idx = torch.arange(T, device=dev) + first_game_index
        one_vs_two = ((idx % 4 == 0) | (idx % 4 == 3))
        counters = torch.zeros(T, dtype=torch.int64, device=dev)
        boards = g.init_boards_batch(T, stream0=self.stream0 + first_game_index, counters=counters)
        cur = torch.zeros(T, dtype=torch.int32, device=dev)
        done = torch.arange(T, device=dev) >= n
        result = torch.zeros(T, dtype=torch.float32, device=dev)
        zero_seed = torch.zeros(T, dtype=torch.int64, device=dev)
        for m in self.mcts:
            m.reset_all_search_trees()                                       # Arena.py:99
        for ply in range(self.max_plies):
            if bool(done.all().item()):
                break
            # seat 0 belongs to contestant 0 in a "one vs two" game, to contestant 1 otherwise; other seats to the other one
            owner = torch.where((cur == 0) == one_vs_two, torch.zeros_like(cur), torch.ones_like(cur))
            canonical = g.canonical_batch(boards, cur)
            actions = torch.zeros(T, dtype=torch.int32, device=dev)
            for c, m in enumerate(self.mcts):
                active = (~done) & (owner == c)
                if not bool(active.any().item()):
                    continue
                full = torch.where(active, torch.ones(T, dtype=torch.uint8, device=dev), torch.full((T,), 2, dtype=torch.uint8, device=dev))
                temp = 1 if self.temp_for_game is None else self.temp_for_game(ply + 1)       # `it` of Arena.py:67-68
                probs, _, _ = m.getActionProb(canonical, temp=temp, full=full)
                ar = torch.arange(probs.shape[1], device=dev)[None, :]              # np.argmax: FIRST index of the maximum
                first_max = torch.where(probs == probs.max(dim=1, keepdim=True).values, ar, probs.shape[1]).min(dim=1).values
                first_max = torch.where(first_max >= probs.shape[1], torch.zeros_like(first_max), first_max)   # all-NaN row -> 0, like np.argmax
                actions = torch.where(active, first_max.to(torch.int32), actions)
            if record is not None:
                record.append((boards.clone(), cur.clone(), actions.clone(), done.clone()))
            nb, ncur = g.next_state_batch(boards, cur, actions, zero_seed, stream0=self.stream0 + first_game_index, counters=counters)
            live = ~done
            boards = torch.where(live[:, None], nb, boards)
            cur = torch.where(live, ncur, cur)
            ended, _, _ = g.game_ended_batch(boards, cur)
            fin = live & (ended != 0).any(dim=1)
            result = torch.where(fin, ended[:, 0], result)
            done = done | fin
        return result[:n], one_vs_two[:n]

    def playGames(self, num, first_game_index=0):
        """-> (oneWon, twoWon, draws) like Arena.playGames (Arena.py:103-140).  first_game_index: this object plays games
        [first_game_index, first_game_index + num) of a match that is dealt out over several ranks (the seating and the random
        streams of a game are functions of its index)"""
        one = two = draws = 0
        for first in range(first_game_index, first_game_index + num, self.T):
            n = min(self.T, first_game_index + num - first)
            res, ovt = self.play_wave(first, n)
            win_first_seat, win_other = res == 1.0, res == -1.0
            one += int(((ovt & win_first_seat) | (~ovt & win_other)).sum().item())
            two += int(((ovt & win_other) | (~ovt & win_first_seat)).sum().item())
            draws += int((~(win_first_seat | win_other)).sum().item())
        return one, two, draws
