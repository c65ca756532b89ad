"""ctypes binding of the CPU ORACLE (oracle/libazg_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (alpha-zero-general_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libazg_oracle.so')

SPLENDOR, SANTORINI, AZUL, MINIVILLES, ABALONE, TLP, BOTANIK, AKROPOLIS, SMALLWORLD = 0, 1, 2, 3, 4, 5, 6, 7, 8
MAXP = 5


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(['make', '-C', HERE, '-s'] + (['-B'] if force else []))
    return LIB_PATH


class Game(C.Structure):
    _fields_ = [('id', C.c_int), ('S', C.c_int), ('A', C.c_int), ('P', C.c_int), ('variant', C.c_int),
                ('rows', C.c_int), ('cols', C.c_int)]


class Rng(C.Structure):
    _fields_ = [('mode', C.c_int), ('seed', C.c_uint64), ('stream', C.c_uint64), ('counter', C.c_uint64),
                ('injected', C.POINTER(C.c_double)), ('n_injected', C.c_size_t), ('pos', C.c_size_t)]


class MctsArgs(C.Structure):
    _fields_ = [('numMCTSSims', C.c_int), ('cpuct', C.c_double), ('fpu', C.c_double), ('universes', C.c_int),
                ('prob_fullMCTS', C.c_double), ('ratio_fullMCTS', C.c_int), ('forced_playouts', C.c_int),
                ('no_mem_optim', C.c_int), ('dirichletAlpha', C.c_double), ('temperature', C.c_double * 3),
                ('numpy2_scalar_typing', C.c_int)]


class EpisodeCfg(C.Structure):
    _fields_ = [('mcts', MctsArgs), ('temp_begin', C.c_double), ('temp_end', C.c_double),
                ('tempThreshold', C.c_double), ('dirichlet_noise', C.c_int), ('max_plies', C.c_int)]


PREDICT_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int8), C.POINTER(C.c_uint8), C.POINTER(C.c_float),
                         C.POINTER(C.c_float))

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i8p, u8p, f32p, f64p, i64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
        L.azo_game_init.argtypes = [C.POINTER(Game), C.c_int, C.c_int]
        L.azo_valid_moves.argtypes = [C.POINTER(Game), i8p, C.c_int, u8p]
        L.azo_make_move.argtypes = [C.POINTER(Game), i8p, C.c_int, C.c_int, C.c_int64, C.POINTER(Rng)]
        L.azo_game_ended.argtypes = [C.POINTER(Game), i8p, C.c_int, f32p]
        L.azo_swap_players.argtypes = [C.POINTER(Game), i8p, C.c_int]
        L.azo_get_round.argtypes = [C.POINTER(Game), i8p]
        L.azo_get_score.argtypes = [C.POINTER(Game), i8p, C.c_int]
        L.azo_init_board.argtypes = [C.POINTER(Game), i8p, C.POINTER(Rng)]
        L.azo_known_start.argtypes = [C.POINTER(Game), i8p, C.c_int, C.c_int]
        L.azo_canonical.argtypes = [C.POINTER(Game), i8p, C.c_int, i8p]
        L.azo_symmetries.argtypes = [C.POINTER(Game), i8p, f32p, u8p, i8p, f32p, u8p, C.c_int]
        L.azo_symmetries_rng.argtypes = [C.POINTER(Game), i8p, f32p, u8p, i8p, f32p, u8p, C.c_int, C.POINTER(Rng)]
        L.azo_rng_u01.restype = C.c_double
        L.azo_rng_u01.argtypes = [C.POINTER(Rng)]
        L.azo_rng_raw.restype = C.c_uint64
        L.azo_rng_raw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.azo_mcts_create.restype = vp
        L.azo_mcts_create.argtypes = [C.POINTER(Game), C.POINTER(MctsArgs), C.c_int]
        L.azo_mcts_destroy.argtypes = [vp]
        L.azo_mcts_reset.argtypes = [vp]
        L.azo_mcts_set_rng.argtypes = [vp, C.POINTER(Rng)]
        L.azo_mcts_num_nodes.restype = C.c_size_t
        L.azo_mcts_num_nodes.argtypes = [vp]
        L.azo_hashnet_predict.argtypes = [vp, i8p, u8p, f32p, f32p]
        L.azo_mcts_get_action_prob.argtypes = [vp, i8p, C.c_double, C.c_int, C.c_double, f64p, vp, vp, f64p, f32p]
        L.azo_mcts_search_begin.argtypes = [vp, i8p, C.c_int, C.c_double, f64p]
        L.azo_mcts_search_done.argtypes = [vp]
        L.azo_mcts_sim_begin.argtypes = [vp]
        L.azo_mcts_leaf_board.restype = C.POINTER(C.c_int8)
        L.azo_mcts_leaf_board.argtypes = [vp]
        L.azo_mcts_leaf_valids.restype = C.POINTER(C.c_uint8)
        L.azo_mcts_leaf_valids.argtypes = [vp]
        L.azo_mcts_sim_finish.argtypes = [vp, f32p, f32p]
        L.azo_mcts_search_end.argtypes = [vp, C.c_double, f64p, f32p]
        L.azo_mcts_node_stats.argtypes = [vp, i8p, i64p, f32p, i64p, f64p, f32p, f32p, C.POINTER(C.c_int)]
        L.azo_mcts_dump_keys.restype = C.c_size_t
        L.azo_mcts_dump_keys.argtypes = [vp, i8p, C.c_size_t]
        L.azo_mcts_counters.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 5
        L.azo_episode_run.argtypes = [C.POINTER(Game), C.POINTER(EpisodeCfg), i8p, C.c_uint64, C.c_uint64, vp, vp,
                                      i8p, f64p, f32p, vp, vp, vp, f32p, i8p]
        L.azo_version.restype = C.c_char_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_args(numMCTSSims=800, cpuct=1.0, fpu=0.0, universes=1, prob_fullMCTS=1.0, ratio_fullMCTS=5,
              forced_playouts=True, no_mem_optim=True, dirichletAlpha=0.0, temperature=(1.0, 1.0, 1.0),
              numpy2_scalar_typing=False, **_ignored):
    a = MctsArgs()
    a.numMCTSSims = numMCTSSims
    a.cpuct = cpuct
    a.fpu = fpu
    a.universes = universes
    a.prob_fullMCTS = prob_fullMCTS
    a.ratio_fullMCTS = ratio_fullMCTS
    a.forced_playouts = int(forced_playouts)
    a.no_mem_optim = int(no_mem_optim)
    a.dirichletAlpha = dirichletAlpha
    for i in range(3):
        a.temperature[i] = temperature[i] if i < len(temperature) else 1.0
    a.numpy2_scalar_typing = int(numpy2_scalar_typing)
    return a


class OracleGame:
    """Game.py-shaped view of the oracle (same method names and argument meaning as the reference's <G>Game)."""

    def __init__(self, game_id, variant=0):
        self.g = Game()
        if lib().azo_game_init(C.byref(self.g), game_id, variant) != 0:
            raise ValueError('bad game/variant')
        self.S, self.A, self.P = self.g.S, self.g.A, self.g.P
        self.num_players = self.P
        self.shape = (5, 5, 3) if game_id == SANTORINI else (9, 9, 4) if game_id == ABALONE else (66, 5, 7) if game_id == BOTANIK else (13, 13, self.g.cols) if game_id == AKROPOLIS else (self.g.rows, self.g.cols)   # Azul: (23, 6)

    def getBoardSize(self):
        return self.shape

    def getActionSize(self):
        return self.A

    def rng(self, seed=0, stream=0, injected=None):
        r = Rng()
        if injected is not None:
            arr = np.ascontiguousarray(injected, dtype=np.float64)
            r._keep = arr
            r.mode = 1
            r.injected = arr.ctypes.data_as(C.POINTER(C.c_double))
            r.n_injected = arr.size
        else:
            r.mode, r.seed, r.stream, r.counter = 0, seed, stream, 0
        return r

    def getInitBoard(self, rng):
        st = np.zeros(self.S, dtype=np.int8)
        lib().azo_init_board(C.byref(self.g), _p(st), C.byref(rng))
        return st.reshape(self.shape)

    def known_start(self, a=0, b=0):
        st = np.zeros(self.S, dtype=np.int8)
        lib().azo_known_start(C.byref(self.g), _p(st), a, b)
        return st

    def getValidMoves(self, board, player):
        b = np.ascontiguousarray(board, dtype=np.int8)
        out = np.zeros(self.A, dtype=np.uint8)
        lib().azo_valid_moves(C.byref(self.g), _p(b), int(player), _p(out))
        return out.astype(bool)

    def getNextState(self, board, player, action, random_seed=0, rng=None):
        b = np.array(board, dtype=np.int8, copy=True, order='C')
        np_ = lib().azo_make_move(C.byref(self.g), _p(b), int(action), int(player), int(random_seed),
                                  C.byref(rng) if rng is not None else None)
        return b, np_

    def getGameEnded(self, board, next_player):
        b = np.ascontiguousarray(board, dtype=np.int8)
        out = np.zeros(self.P, dtype=np.float32)
        lib().azo_game_ended(C.byref(self.g), _p(b), int(next_player), _p(out))
        return out

    def getScore(self, board, player):
        return lib().azo_get_score(C.byref(self.g), _p(np.ascontiguousarray(board, dtype=np.int8)), int(player))

    def getRound(self, board):
        return lib().azo_get_round(C.byref(self.g), _p(np.ascontiguousarray(board, dtype=np.int8)))

    def getCanonicalForm(self, board, player):
        b = np.ascontiguousarray(board, dtype=np.int8)
        out = np.empty_like(b)
        lib().azo_canonical(C.byref(self.g), _p(b), int(player), _p(out))
        return out

    def getSymmetries(self, board, pi, valids, max_sym=16, rng=None):
        b = np.ascontiguousarray(board, dtype=np.int8)
        pi = np.ascontiguousarray(pi, dtype=np.float32)
        va = np.ascontiguousarray(valids, dtype=np.uint8)
        os_ = np.zeros((max_sym, self.S), dtype=np.int8)
        op = np.zeros((max_sym, self.A), dtype=np.float32)
        ov = np.zeros((max_sym, self.A), dtype=np.uint8)
        if rng is not None:
            k = lib().azo_symmetries_rng(C.byref(self.g), _p(b), _p(pi), _p(va), _p(os_), _p(op), _p(ov), max_sym, C.byref(rng))
        else:
            k = lib().azo_symmetries(C.byref(self.g), _p(b), _p(pi), _p(va), _p(os_), _p(op), _p(ov), max_sym)
        return [(os_[i].reshape(self.shape), op[i], ov[i].astype(bool)) for i in range(k)]

    def stringRepresentation(self, board):
        return np.ascontiguousarray(board, dtype=np.int8).tobytes()


class OracleMCTS:
    """MCTS.py-shaped view of the oracle.  predict: None -> built-in C hash-net; else callable(board, valids)->(pi, v)."""

    def __init__(self, game, args, dirichlet_noise=False, predict=None):
        self.game = game
        self.args = args if isinstance(args, MctsArgs) else make_args(**dict(args))
        self.h = lib().azo_mcts_create(C.byref(game.g), C.byref(self.args), int(dirichlet_noise))
        self._predict = predict
        if predict is None:
            self._cb = C.cast(lib().azo_hashnet_predict, C.c_void_p)
            self._ctx = C.cast(C.pointer(game.g), C.c_void_p)
        else:
            S, A, P = game.S, game.A, game.P

            def cb(ctx, board, valids, pi, v):
                b = np.ctypeslib.as_array(board, shape=(S,)).reshape(game.shape)
                va = np.ctypeslib.as_array(valids, shape=(A,)).astype(bool)
                ppi, vv = predict(b, va)
                np.ctypeslib.as_array(pi, shape=(A,))[:] = np.asarray(ppi, dtype=np.float32)
                np.ctypeslib.as_array(v, shape=(P,))[:] = np.asarray(vv, dtype=np.float32)

            self._cbobj = PREDICT_FN(cb)
            self._cb = C.cast(self._cbobj, C.c_void_p)
            self._ctx = None

    def __del__(self):
        if getattr(self, 'h', None):
            lib().azo_mcts_destroy(self.h)
            self.h = None

    def set_rng(self, rng):
        """random source of the env steps inside the search (games that roll dice in make_move: Minivilles); borrowed"""
        self._rng = rng
        lib().azo_mcts_set_rng(self.h, C.byref(rng) if rng is not None else None)

    def getActionProb(self, canonicalBoard, temp=1, force_full_search=False, u_full=0.0, dir_noise=None):
        b = np.ascontiguousarray(canonicalBoard, dtype=np.int8)
        probs = np.zeros(self.game.A, dtype=np.float64)
        q = np.zeros(self.game.P, dtype=np.float32)
        dn = None if dir_noise is None else np.ascontiguousarray(dir_noise, dtype=np.float64)
        full = lib().azo_mcts_get_action_prob(self.h, _p(b), float(temp), int(force_full_search), float(u_full),
                                              _p(dn), self._cb, self._ctx, _p(probs), _p(q))
        if full < 0:
            raise RuntimeError('oracle mcts failed')
        return probs, q, bool(full)

    def num_nodes(self):
        return lib().azo_mcts_num_nodes(self.h)

    def node(self, state):
        A, P = self.game.A, self.game.P
        b = np.ascontiguousarray(state, dtype=np.int8)
        Ns = np.zeros(1, dtype=np.int64)
        Qs = np.zeros(1, dtype=np.float32)
        Nsa = np.zeros(A, dtype=np.int64)
        Qsa = np.zeros(A, dtype=np.float64)
        Ps = np.zeros(A, dtype=np.float32)
        Es = np.zeros(P, dtype=np.float32)
        hp = C.c_int(0)
        ok = lib().azo_mcts_node_stats(self.h, _p(b), _p(Ns), _p(Qs), _p(Nsa), _p(Qsa), _p(Ps), _p(Es), C.byref(hp))
        if not ok:
            return None
        return dict(Ns=int(Ns[0]), Qs=Qs[0], Nsa=Nsa, Qsa=Qsa, Ps=Ps, Es=Es, has_policy=bool(hp.value))

    def keys(self):
        n = self.num_nodes()
        out = np.zeros((max(n, 1), self.game.S), dtype=np.int8)
        lib().azo_mcts_dump_keys(self.h, _p(out), n)
        return out[:n]

    def counters(self):
        vals = [C.c_uint64(0) for _ in range(5)]
        lib().azo_mcts_counters(self.h, *[C.byref(v) for v in vals])
        return dict(zip(['sims', 'levels', 'expansions', 'sum_valid_visited', 'terminal_hits'],
                        [int(v.value) for v in vals]))

    # resumable API (batched leaf evaluation)
    def search_begin(self, canonical, force_full_search=True, u_full=0.0):
        b = np.ascontiguousarray(canonical, dtype=np.int8)
        lib().azo_mcts_search_begin(self.h, _p(b), int(force_full_search), float(u_full), None)

    def search_done(self):
        return bool(lib().azo_mcts_search_done(self.h))

    def sim_begin(self):
        return lib().azo_mcts_sim_begin(self.h)

    def leaf(self):
        S, A = self.game.S, self.game.A
        b = np.ctypeslib.as_array(lib().azo_mcts_leaf_board(self.h), shape=(S,))
        v = np.ctypeslib.as_array(lib().azo_mcts_leaf_valids(self.h), shape=(A,))
        return b, v

    def sim_finish(self, pi, v):
        pi = np.ascontiguousarray(pi, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        lib().azo_mcts_sim_finish(self.h, _p(pi), _p(v))

    def search_end(self, temp=1.0):
        probs = np.zeros(self.game.A, dtype=np.float64)
        q = np.zeros(self.game.P, dtype=np.float32)
        full = lib().azo_mcts_search_end(self.h, float(temp), _p(probs), _p(q))
        return probs, q, bool(full)


def hashnet_predict(game, board, valids):
    b = np.ascontiguousarray(board, dtype=np.int8)
    va = np.ascontiguousarray(valids, dtype=np.uint8)
    pi = np.zeros(game.A, dtype=np.float32)
    v = np.zeros(game.P, dtype=np.float32)
    lib().azo_hashnet_predict(C.cast(C.pointer(game.g), C.c_void_p), _p(b), _p(va), _p(pi), _p(v))
    return pi, v


def run_episode(game, mcts_args, init_board=None, seed=0, stream=0, temp=(1.0, 1.0), tempThreshold=10.0,
                max_plies=512, predict=None):
    """Coach.executeEpisode on the oracle with the counter-based RNG.  Returns dict of per-ply records."""
    cfg = EpisodeCfg()
    cfg.mcts = mcts_args
    cfg.temp_begin, cfg.temp_end, cfg.tempThreshold = temp[0], temp[1], tempThreshold
    cfg.dirichlet_noise = 0
    cfg.max_plies = max_plies
    S, A, P = game.S, game.A, game.P
    canon = np.zeros((max_plies, S), dtype=np.int8)
    pi = np.zeros((max_plies, A), dtype=np.float64)
    q = np.zeros((max_plies, P), dtype=np.float32)
    act = np.zeros(max_plies, dtype=np.int32)
    ply_player = np.zeros(max_plies, dtype=np.int32)
    full = np.zeros(max_plies, dtype=np.int32)
    result = np.zeros(P, dtype=np.float32)
    final = np.zeros(S, dtype=np.int8)
    ib = None if init_board is None else np.ascontiguousarray(init_board, dtype=np.int8)
    if predict is None:
        cb = C.cast(lib().azo_hashnet_predict, C.c_void_p)
        ctx = C.cast(C.pointer(game.g), C.c_void_p)
    else:
        raise NotImplementedError
    n = lib().azo_episode_run(C.byref(game.g), C.byref(cfg), _p(ib), seed, stream, cb, ctx, _p(canon), _p(pi), _p(q),
                              _p(act), _p(ply_player), _p(full), _p(result), _p(final))
    if n < 0:
        raise RuntimeError('oracle episode failed: %d' % n)
    return dict(plies=n, canonical=canon[:n], pi=pi[:n], q=q[:n], action=act[:n], player=ply_player[:n],
                full=full[:n], result=result, final_board=final)
