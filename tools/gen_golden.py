"""Generate golden vectors from the REFERENCE (imported live, pure-Python mode) -> tests/golden/*.npz.

Build-container only (needs /root/reference).  The outputs are data: inputs and the reference's outputs.
    python tools/gen_golden.py            # the small committed set (tests/golden/)
    python tools/gen_golden.py --big DIR  # a much larger uncommitted set for tools/check_oracle_vs_ref.py

Fixture families:
  G1 env_<variant>.npz   seeded random-play transitions: state, player, valid mask, action, random_seed, next state,
                         next player, game_ended, scores, round, canonical form; random_seed==0 moves are recorded with
                         the uniforms that were injected into the reference's np.random.random().
  G2 (inside G1)         captured getInitBoard() outputs (RNG dependent in the reference).
  G3 mcts_<variant>.npz  MCTS traces with the integer hash-net (Appendix C.3): root statistics, returned probs/q,
                         whole-tree digest, for several (sims, args) incl. a multi-move sequence with tree reuse and
                         the periodic clean-up.
  G5 sym_<variant>.npz   getSymmetries outputs.
"""
import argparse
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402

GOLDEN = os.path.join(HERE, '..', 'tests', 'golden')

VARIANTS = {
    # name: (loader kwargs, game module key, game class name)
    'splendor2': (dict(splendor_players=2), 'SplendorGame', 'SplendorGame'),
    'splendor3': (dict(splendor_players=3), 'SplendorGame', 'SplendorGame'),
    'splendor4': (dict(splendor_players=4), 'SplendorGame', 'SplendorGame'),
    'santorini1': (dict(santorini_gods=1), 'SantoriniGame', 'SantoriniGame'),
    'santorini11': (dict(santorini_gods=11), 'SantoriniGame', 'SantoriniGame'),
    'azul': (dict(), 'AzulGame', 'AzulGame'),
}

MCTS_ARGS = {
    'splendor2': dict(cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True),
    'splendor3': dict(cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True),
    'splendor4': dict(cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True),
    'santorini1': dict(cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True),
    'santorini11': dict(cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True),
    'azul': dict(cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True),
}


class _PatchedRandom:
    """Feeds the reference's np.random.random() from a recorded uniform list."""

    def __init__(self):
        self.queue = []
        self.used = []
        self.orig = np.random.random

    def __enter__(self):
        def fake():
            v = self.queue.pop(0)
            self.used.append(v)
            return v
        np.random.random = fake
        return self

    def __exit__(self, *a):
        np.random.random = self.orig


def gen_env(name, n_traj, rng, max_plies=400):
    kw, modkey, cls = VARIANTS[name]
    m = H.load_reference(**kw)
    game = getattr(m[modkey], cls)()
    P = game.num_players
    A = game.getActionSize()
    rec = {k: [] for k in ('state', 'player', 'valid', 'action', 'seed', 'next_state', 'next_player', 'ended',
                           'score', 'round', 'canonical', 'uniforms', 'traj')}
    inits = []
    for t in range(n_traj):
        np.random.seed(1000 + t)
        board = game.getInitBoard().copy()
        inits.append(board.copy())
        player = 0
        for ply in range(max_plies):
            valid = game.getValidMoves(board, player).copy()
            idx = np.flatnonzero(valid)
            if len(idx) == 0:
                break
            if name.startswith('splendor'):
                # bias towards buy / reserve so that decks run out and nobles / end-of-game paths are reached
                buyres = idx[idx < 30]
                a = int(rng.choice(buyres)) if (len(buyres) and rng.random() < 0.6) else int(rng.choice(idx))
            else:
                a = int(rng.choice(idx))
            r = rng.random()
            if r < 0.25:
                seed = 0
            elif r < 0.35:
                seed = -1
            else:
                seed = H.MAGIC_SEEDS[int(rng.integers(8))]
            nu = 24 if name == 'azul' else 4       # Azul's new-round refill draws up to 20 tiles
            us = [float(x) for x in rng.random(nu)]
            with _PatchedRandom() as pr:
                pr.queue = list(us)
                nb, npl = game.getNextState(board, player, a, random_seed=seed)
                used = list(pr.used)
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy())
            rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8)))
            rec['action'].append(a)
            rec['seed'].append(seed)
            rec['next_state'].append(nb.reshape(-1).copy())
            rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(P)])
            rec['round'].append(int(game.getRound(nb)))
            rec['canonical'].append(game.getCanonicalForm(nb, npl).reshape(-1).copy())
            rec['uniforms'].append((used + [0.5] * nu)[:nu])
            rec['traj'].append(t)
            board, player = nb, npl
            if ended.any():
                break
    out = dict(
        state=np.array(rec['state'], dtype=np.int8), player=np.array(rec['player'], dtype=np.int8),
        valid=np.array(rec['valid'], dtype=np.uint8), action=np.array(rec['action'], dtype=np.int16),
        seed=np.array(rec['seed'], dtype=np.int32), next_state=np.array(rec['next_state'], dtype=np.int8),
        next_player=np.array(rec['next_player'], dtype=np.int8), ended=np.array(rec['ended'], dtype=np.float32),
        score=np.array(rec['score'], dtype=np.int16), round=np.array(rec['round'], dtype=np.int16),
        canonical=np.array(rec['canonical'], dtype=np.int8), uniforms=np.array(rec['uniforms'], dtype=np.float64),
        traj=np.array(rec['traj'], dtype=np.int16), init_boards=np.array([b.reshape(-1) for b in inits], dtype=np.int8),
        shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(P))
    return out, m, game


def tree_digest(mc, A):
    """Digest over every node: key bytes | Ns | Nsa(i64) | Qsa(f64) | Ps(f32) | Qs(f32) | Es(f32)."""
    h = hashlib.sha256()
    for key in sorted(mc.nodes_data.keys()):
        Es, Vs, Ps, Ns, Qsa, Nsa, r, Qs = mc.nodes_data[key]
        h.update(key)
        h.update(np.asarray(Es, dtype=np.float32).tobytes())
        if Ps is not None:
            h.update(np.int64(Ns).tobytes())
            h.update(np.asarray(Nsa, dtype=np.int64).tobytes())
            h.update(np.asarray(Qsa, dtype=np.float64).tobytes())
            h.update(np.asarray(Ps, dtype=np.float32).tobytes())
            h.update(np.float32(Qs).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def gen_mcts(name, env, m, game, rng, sims_list, n_roots, seq_moves, typed):
    M = m['MCTS']
    if typed:
        H.enable_numba_typing(M)
    P, A = game.num_players, game.getActionSize()
    shape = tuple(env['shape'])
    cases = []
    # roots: init boards + mid-game canonical states
    mid = rng.choice(len(env['canonical']), size=n_roots, replace=False)
    roots = [env['init_boards'][0].copy()] + [env['canonical'][i].copy() for i in mid if not env['ended'][i].any()]
    variants = [dict(), dict(universes=1), dict(forced_playouts=False, fpu=0.0), dict(fpu=-0.1, universes=0)]
    for ri, root in enumerate(roots):
        for sims in sims_list:
            for vi, var in enumerate(variants if ri < 2 else variants[:1]):
                kw = dict(MCTS_ARGS[name])
                kw.update(var)
                args = H.mcts_args(m['utils'], numMCTSSims=sims, **kw)
                mc = M.MCTS(game, H.HashNet(P), args)
                board = root.reshape(shape)
                probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
                nd = mc.nodes_data[board.tobytes()]
                cases.append(dict(root=root, sims=sims, cpuct=kw['cpuct'], fpu=kw['fpu'], universes=kw['universes'],
                                  forced=int(kw['forced_playouts']), Ns=nd[3], Qs=np.float32(nd[7]),
                                  Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                                  Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                                  q=np.asarray(q, dtype=np.float32), nodes=len(mc.nodes_data),
                                  digest=tree_digest(mc, A)))
    # multi-move sequence with tree reuse + periodic clean-up (no_mem_optim=False)
    seq = []
    kw = dict(MCTS_ARGS[name])
    args = H.mcts_args(m['utils'], numMCTSSims=sims_list[0] * 2, no_mem_optim=False, prob_fullMCTS=0.0, **kw)
    mc = M.MCTS(game, H.HashNet(P), args)
    board = env['init_boards'][1].reshape(shape).copy()
    player = 0
    for ply in range(seq_moves):
        canon = game.getCanonicalForm(board, player)
        probs, q, full = mc.getActionProb(canon, temp=1, force_full_search=(ply % 3 != 2))
        probs = np.asarray(probs, dtype=np.float64)
        a = int(np.argmax(probs))
        nd = mc.nodes_data[canon.tobytes()]
        seq.append(dict(canon=canon.reshape(-1).copy(), board=board.reshape(-1).copy(), player=player, probs=probs,
                        q=np.asarray(q, dtype=np.float32), Ns=nd[3], Nsa=np.asarray(nd[5], dtype=np.int64),
                        action=a, nodes=len(mc.nodes_data), full=int(full), digest=tree_digest(mc, A)))
        board, player = game.getNextState(board, player, a, random_seed=H.MAGIC_SEEDS[ply % 8])
        board = board.copy()
        if game.getGameEnded(board, player).any():
            break
    out = {}
    for k in cases[0]:
        out['case_' + k] = np.array([c[k] for c in cases])
    for k in seq[0]:
        out['seq_' + k] = np.array([s[k] for s in seq])
    out['seq_sims'] = np.array(sims_list[0] * 2)
    out['typed'] = np.array(int(typed))
    return out


def gen_sym(name, env, game, rng, n):
    A = game.getActionSize()
    shape = tuple(env['shape'])
    sel = rng.choice(len(env['canonical']), size=n, replace=False)
    states, pis, valids, o_states, o_pi, o_valid, counts = [], [], [], [], [], [], []
    for i in sel:
        st = env['canonical'][i].reshape(shape).copy()
        va = game.getValidMoves(st, 0).copy()
        pi = (rng.random(A) * va).astype(np.float32)
        pi /= max(pi.sum(), 1e-9)
        syms = game.getSymmetries(st, pi, va)
        states.append(st.reshape(-1)); pis.append(pi); valids.append(va.astype(np.uint8))
        counts.append(len(syms))
        pad = (128 if len(syms) > 24 else 24) - len(syms)
        o_states.append(np.array([s[0].reshape(-1) for s in syms] + [np.zeros(st.size, np.int8)] * pad, dtype=np.int8))
        o_pi.append(np.array([s[1] for s in syms] + [np.zeros(A, np.float32)] * pad, dtype=np.float32))
        o_valid.append(np.array([np.asarray(s[2]).astype(np.uint8) for s in syms] + [np.zeros(A, np.uint8)] * pad))
    return dict(state=np.array(states), pi=np.array(pis), valid=np.array(valids), count=np.array(counts),
                out_state=np.array(o_states), out_pi=np.array(o_pi), out_valid=np.array(o_valid))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--big', default=None)
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    big = a.big is not None
    outdir = a.big if big else GOLDEN
    os.makedirs(outdir, exist_ok=True)
    for name in VARIANTS:
        if a.only and name != a.only:
            continue
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        n_traj = (60 if big else 6) if name != 'splendor3' else (20 if big else 2)
        env, m, game = gen_env(name, n_traj, rng)
        np.savez_compressed(os.path.join(outdir, 'env_%s.npz' % name), **env)
        print(name, 'env transitions', len(env['state']))
        sym = gen_sym(name, env, game, rng, (40 if big else 6) if name != 'azul' else (6 if big else 2))
        np.savez_compressed(os.path.join(outdir, 'sym_%s.npz' % name), **sym)
        if name in ('splendor3',):
            H.cleanup()
            continue
        for typed in (0, 1):
            mc = gen_mcts(name, env, m, game, rng, sims_list=[25, 200] if not big else [25, 200, 800],
                          n_roots=3 if not big else 8, seq_moves=30 if not big else 60, typed=typed)
            np.savez_compressed(os.path.join(outdir, 'mcts_%s_%s.npz' % (name, 'numba' if typed else 'numpy2')), **mc)
            print(name, 'mcts cases', len(mc['case_sims']), 'seq', len(mc['seq_action']), 'typed', typed)
            if typed:  # need a fresh module for the untyped run next variant
                pass
        H.cleanup()


if __name__ == '__main__':
    main()
