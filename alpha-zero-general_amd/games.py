"""Game.py-compatible env plugins backed by the batched HIP env kernels (azg_env_* in include/azg.h).

Same method names, argument meaning and return types as the reference's adaptors (splendor/SplendorGame.py:16-60,
santorini/SantoriniGame.py:16-60, Game.py:14-162).  Every method also has a `*_batch` twin that takes and returns
torch CUDA tensors for n boards at once -- that is the form the engine itself uses."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HipGame:
    GAME_ID = None

    def __init__(self, variant, device='cuda:0', rng_seed=0):
        if not torch.cuda.is_available():
            raise _lib.AzgError('no GPU visible: the engine has no CPU fallback')
        self.variant = variant
        self.device = torch.device(device)
        self.S, self.A, self.P, self.rows, self.cols = _lib.game_info(self.GAME_ID, variant)
        self.num_players = self.P
        self.rng_seed = rng_seed
        self._stream_ctr = 0
        self._draw = torch.zeros(1, dtype=torch.int64, device=self.device)

    # ---- batch API (torch CUDA tensors) ----
    def valid_moves_batch(self, boards, players):
        n = boards.shape[0]
        out = torch.empty((n, self.A), dtype=torch.uint8, device=self.device)
        check(lib().azg_env_valid_moves(self.GAME_ID, self.variant, _ptr(boards), _ptr(players), n, _ptr(out),
                                        _stream()))
        return out

    def next_state_batch(self, boards, players, actions, seeds, stream0=0, counters=None):
        n = boards.shape[0]
        out = torch.empty((n, self.S), dtype=torch.int8, device=self.device)
        nxt = torch.empty((n,), dtype=torch.int32, device=self.device)
        check(lib().azg_env_next_state(self.GAME_ID, self.variant, _ptr(boards), _ptr(players), _ptr(actions),
                                       _ptr(seeds), n, _ptr(out), _ptr(nxt), self.rng_seed, stream0, _ptr(counters),
                                       _stream()))
        return out, nxt

    def game_ended_batch(self, boards, next_players):
        n = boards.shape[0]
        ended = torch.empty((n, self.P), dtype=torch.float32, device=self.device)
        scores = torch.empty((n, self.P), dtype=torch.int32, device=self.device)
        rnd = torch.empty((n,), dtype=torch.int32, device=self.device)
        check(lib().azg_env_game_ended(self.GAME_ID, self.variant, _ptr(boards), _ptr(next_players), n, _ptr(ended),
                                       _ptr(scores), _ptr(rnd), _stream()))
        return ended, scores, rnd

    def canonical_batch(self, boards, players):
        n = boards.shape[0]
        out = torch.empty((n, self.S), dtype=torch.int8, device=self.device)
        check(lib().azg_env_canonical(self.GAME_ID, self.variant, _ptr(boards), _ptr(players), n, _ptr(out), _stream()))
        return out

    def init_boards_batch(self, n, stream0=0, counters=None):
        """Board.init_game for n games on RNG streams stream0..stream0+n-1; `counters` (u64[n] tensor) receives the number
        of uniforms each stream consumed, so next_state_batch(..., counters=counters) continues the same streams"""
        out = torch.empty((n, self.S), dtype=torch.int8, device=self.device)
        check(lib().azg_env_init_boards(self.GAME_ID, self.variant, n, _ptr(out), self.rng_seed, stream0, _ptr(counters),
                                        _stream()))
        return out

    def max_symmetries(self):
        return {0: 10 + 2 * self.P, 1: 8, 2: 120, 3: 1, 4: 12, 5: 2 * self.P + 1, 6: 14, 7: 6, 8: 3}[self.GAME_ID]    # Splendor, Santorini, Azul, Minivilles, Abalone, TLP, Botanik, Akropolis, Smallworld

    def symmetries_batch(self, boards, pi, valids, max_sym=None, rng_seed=None, stream0=0):
        """getSymmetries for n (board int8[n,S], pi f32[n,A], valids u8[n,A]) triples on device ->
        (boards int8[n,K,S], pi f32[n,K,A], valids u8[n,K,A], count i32[n]), K = max_sym; rows >= count[t] are unspecified.
        Games whose symmetries are random (The Little Prince shuffles) draw triple t from stream (rng_seed, stream0 + t)."""
        n = boards.shape[0]
        K = max_sym or self.max_symmetries()
        boards = boards.reshape(n, -1)
        valids = valids if valids.dtype == torch.uint8 else valids.to(torch.uint8)
        assert boards.dtype == torch.int8 and pi.dtype == torch.float32
        assert boards.is_contiguous() and pi.is_contiguous() and valids.is_contiguous()
        ob = torch.empty((n, K, self.S), dtype=torch.int8, device=self.device)
        op = torch.empty((n, K, self.A), dtype=torch.float32, device=self.device)
        ov = torch.empty((n, K, self.A), dtype=torch.uint8, device=self.device)
        cnt = torch.zeros((n,), dtype=torch.int32, device=self.device)
        check(lib().azg_env_symmetries_ex(self.GAME_ID, self.variant, _ptr(boards), _ptr(pi), _ptr(valids), n, K, _ptr(ob), _ptr(op),
                                          _ptr(ov), _ptr(cnt), self.rng_seed if rng_seed is None else rng_seed, stream0, _stream()))
        return ob, op, ov, cnt

    # ---- Game.py API (numpy in / numpy out, one board) ----
    def _dev(self, board):
        return torch.from_numpy(np.ascontiguousarray(board, dtype=np.int8).reshape(1, self.S)).to(self.device)

    def _i32(self, v):
        return torch.tensor([int(v)], dtype=torch.int32, device=self.device)

    def getBoardSize(self):
        return ((5, 5, 3) if self.GAME_ID == _lib.SANTORINI else (9, 9, 4) if self.GAME_ID == _lib.ABALONE else (66, 5, 7) if self.GAME_ID == _lib.BOTANIK
                else (13, 13, self.cols) if self.GAME_ID == _lib.AKROPOLIS else (self.rows, self.cols))

    def getActionSize(self):
        return self.A

    def getNumberOfPlayers(self):
        return self.P

    def getInitBoard(self):
        self._stream_ctr += 1
        return self.init_boards_batch(1, stream0=self._stream_ctr)[0].cpu().numpy().reshape(self.getBoardSize())

    def getNextState(self, board, player, action, random_seed=0):
        seeds = torch.tensor([int(random_seed)], dtype=torch.int64, device=self.device)
        self._stream_ctr += 1
        out, nxt = self.next_state_batch(self._dev(board), self._i32(player), self._i32(action), seeds,
                                         stream0=(1 << 40) + self._stream_ctr)
        return out[0].cpu().numpy().reshape(self.getBoardSize()), int(nxt[0].item())

    def getValidMoves(self, board, player):
        return self.valid_moves_batch(self._dev(board), self._i32(player))[0].cpu().numpy().astype(bool)

    def getGameEnded(self, board, next_player):
        return self.game_ended_batch(self._dev(board), self._i32(next_player))[0][0].cpu().numpy()

    def getScore(self, board, player):
        return int(self.game_ended_batch(self._dev(board), self._i32(0))[1][0, player].item())

    def getRound(self, board):
        return int(self.game_ended_batch(self._dev(board), self._i32(0))[2][0].item())

    def getCanonicalForm(self, board, player):
        if player == 0:
            return board
        return self.canonical_batch(self._dev(board), self._i32(player))[0].cpu().numpy().reshape(self.getBoardSize())

    def getSymmetries(self, board, pi, valid_actions):
        """Game.getSymmetries (Game.py:96-109): list of (board, pi, valids) forms, identity first"""
        import numpy as np
        b = self._dev(board).reshape(1, -1)
        p = torch.as_tensor(np.asarray(pi, dtype=np.float32)).reshape(1, -1).to(self.device)
        v = torch.as_tensor(np.asarray(valid_actions).astype(np.uint8)).reshape(1, -1).to(self.device)
        self._stream_ctr += 1
        ob, op, ov, cnt = self.symmetries_batch(b, p, v, stream0=(1 << 41) + self._stream_ctr)
        k = int(cnt[0])
        shape = tuple(self.getBoardSize())
        ob, op, ov = ob[0, :k].cpu().numpy(), op[0, :k].cpu().numpy(), ov[0, :k].cpu().numpy().astype(bool)
        return [(ob[i].reshape(shape), op[i], ov[i]) for i in range(k)]

    def stringRepresentation(self, board):
        return np.ascontiguousarray(board, dtype=np.int8).tobytes()


class SplendorGame(HipGame):
    GAME_ID = _lib.SPLENDOR

    def __init__(self, num_players=2, **kw):
        super().__init__(num_players, **kw)


class SantoriniGame(HipGame):
    GAME_ID = _lib.SANTORINI

    def __init__(self, nb_gods=11, **kw):
        super().__init__(nb_gods, **kw)


class AzulGame(HipGame):
    GAME_ID = _lib.AZUL

    def __init__(self, **kw):
        super().__init__(2, **kw)


class MinivillesGame(HipGame):
    """minivilles/MinivillesGame.py (NUMBER_PLAYERS 2..4).  The env step rolls dice whatever random_seed says
    (MinivillesLogicNumba.py:232-242): every getNextState / getInitBoard draws from the engine's counter RNG stream."""
    GAME_ID = _lib.MINIVILLES

    def __init__(self, num_players=2, **kw):
        super().__init__(num_players, **kw)


class AbaloneGame(HipGame):
    """abalone/AbaloneGame.py (2 players, Belgian-Daisy layout, no dynamic komi: the shipped constants)"""
    GAME_ID = _lib.ABALONE

    def __init__(self, **kw):
        super().__init__(1, **kw)


class TLPGame(HipGame):
    """thelittleprince/TLPGame.py (NUMBER_PLAYERS 3, 4 or 5).  The market refill inside the env step
    and getSymmetries draw true randomness (TLPLogicNumba.py:366-392, 177-272): both use the engine's counter RNG streams."""
    GAME_ID = _lib.TLP

    def __init__(self, num_players=3, **kw):
        super().__init__(num_players, **kw)


class BotanikGame(HipGame):
    """botanik/BotanikGame.py (2 players, MACHINE_SIZE = 7: the shipped constants).  Every card drawn inside the env step takes a
    uniform of the engine's counter RNG stream (BotanikLogicNumba.py:414-438)."""
    GAME_ID = _lib.BOTANIK

    def __init__(self, **kw):
        super().__init__(2, **kw)


class AkropolisGame(HipGame):
    """akropolis/AkropolisGame.py (N_PLAYERS 2 -- the shipped constant -- 3 or 4; 13 x 13 cities, N + 2 tiles on the construction site)"""
    GAME_ID = _lib.AKROPOLIS

    def __init__(self, num_players=2, **kw):
        super().__init__(num_players, **kw)


class SmallworldGame(HipGame):
    """smallworld/SmallworldGame.py (NUMBER_PLAYERS 2 -- the shipped constant -- 3 or 4, each with its own map).  getSymmetries draws two random
    score offsets (SmallworldLogicNumba.py:281-299) from the engine's counter RNG streams."""
    GAME_ID = _lib.SMALLWORLD

    def __init__(self, num_players=2, **kw):
        super().__init__(num_players, **kw)


def import_game(name, **kw):
    """GameSwitcher.import_game equivalent (GameSwitcher.py:15-24) for the games on the hot path."""
    if name == 'splendor':
        return SplendorGame(**kw)
    if name == 'santorini':
        return SantoriniGame(**kw)
    if name == 'azul':
        return AzulGame(**kw)
    if name == 'minivilles':
        return MinivillesGame(**kw)
    if name == 'abalone':
        return AbaloneGame(**kw)
    if name == 'thelittleprince':
        return TLPGame(**kw)
    if name == 'botanik':
        return BotanikGame(**kw)
    if name == 'akropolis':
        return AkropolisGame(**kw)
    if name == 'smallworld':
        return SmallworldGame(**kw)
    raise ValueError('game %r is not on the accelerated path' % name)
