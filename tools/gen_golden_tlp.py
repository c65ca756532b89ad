"""Golden vectors for The Little Prince plugin from the REFERENCE (imported live, pure-Python mode).
Build-container only:  python tools/gen_golden_tlp.py

The market refill inside make_move (TLPLogicNumba.py:366-392) and get_symmetries (:177-272, np.random.shuffle) draw from the
reference's global RNG, in MCTS simulations too; it is replaced by the counter stream of the engine's RNG contract
(tools/refshim/harness.py CounterRandom: random / randint / shuffle) so that every uniform consumed is known.

  env_tlp<n>.npz         random-play transitions (state, player, valid mask, action, next state, next player, ended, score, round,
                         canonical form) + the uniforms each step consumed + captured init boards with theirs
  sym_tlp<n>.npz         get_symmetries of canonical states on stream (seed, index): forms kept after de-duplication, draws consumed
  mcts_tlp3_numba.npz    MCTS.getActionProb traces (hash-net, Numba operand typing), the refills inside the search drawn from a
                         recorded stream: root statistics, probs, q, node count, whole-tree digest, draws consumed
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402
from gen_golden import GOLDEN, tree_digest  # noqa: E402

MAXU = 6          # a step consumes 0 or 1 + n uniforms (the refill)
SYM_SEED = 606


def gen_env(n, n_traj, seed):
    m = H.load_reference(tlp_players=n)
    keys = ('state', 'player', 'valid', 'action', 'next_state', 'next_player', 'ended', 'score', 'round', 'canonical', 'uniforms',
            'n_uniforms', 'traj')
    rec = {k: [] for k in keys}
    inits, init_u = [], []
    rng = np.random.default_rng(seed)
    with H.CounterRandom(seed=777, stream=0):
        game = m['TLPGame'].TLPGame()                          # the constructor fills a market too
    A = n * n
    for t in range(n_traj):
        with H.CounterRandom(seed=2000 + n, stream=t) as cr:
            board = game.getInitBoard().copy()
            iu = list(cr.used)
        inits.append(board.copy()); init_u.append((iu + [0.5] * MAXU)[:MAXU])
        player, ctr = 0, len(iu)
        for ply in range(200):
            valid = game.getValidMoves(board, player).copy()
            a = int(rng.choice(np.flatnonzero(valid)))
            with H.CounterRandom(seed=2000 + n, stream=t, counter=ctr) as cr:
                nb, npl = game.getNextState(board, player, a, random_seed=int(rng.integers(0, 3)))
                used, ctr = list(cr.used), cr.counter
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy()); rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8))); rec['action'].append(a)
            rec['next_state'].append(nb.reshape(-1).copy()); rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(n)])
            rec['round'].append(int(game.getRound(nb)))
            rec['canonical'].append(game.getCanonicalForm(nb, npl).reshape(-1).copy())
            assert len(used) <= MAXU
            rec['uniforms'].append((used + [0.5] * MAXU)[:MAXU]); rec['n_uniforms'].append(len(used))
            rec['traj'].append(t)
            board, player = nb, npl
            if ended.any():
                break
    out = dict(state=np.array(rec['state'], dtype=np.int8), player=np.array(rec['player'], dtype=np.int8),
               valid=np.array(rec['valid'], dtype=np.uint8), action=np.array(rec['action'], dtype=np.int16),
               next_state=np.array(rec['next_state'], dtype=np.int8), next_player=np.array(rec['next_player'], dtype=np.int8),
               ended=np.array(rec['ended'], dtype=np.float32), score=np.array(rec['score'], dtype=np.int16),
               round=np.array(rec['round'], dtype=np.int16), canonical=np.array(rec['canonical'], dtype=np.int8),
               uniforms=np.array(rec['uniforms'], dtype=np.float64), n_uniforms=np.array(rec['n_uniforms'], dtype=np.int8),
               traj=np.array(rec['traj'], dtype=np.int16), init_boards=np.array([b.reshape(-1) for b in inits], dtype=np.int8),
               init_uniforms=np.array(init_u, dtype=np.float64), shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(n))
    return out, m, game


def gen_sym(env, game, n, every):
    """get_symmetries on every `every`-th canonical state: pi = a random policy over the valid moves of player 0"""
    A, S, K = n * n, env['state'].shape[1], 2 * n + 1
    rng = np.random.default_rng(77 + n)
    rows = np.arange(0, len(env['canonical']), every)
    states, pis, vals, cnt, draws = [], [], [], [], []
    out_s = np.zeros((len(rows), K, S), dtype=np.int8)
    out_p = np.zeros((len(rows), K, A), dtype=np.float32)
    out_v = np.zeros((len(rows), K, A), dtype=np.uint8)
    for j, i in enumerate(rows):
        b = env['canonical'][i].reshape(tuple(env['shape'])).copy()
        v = game.getValidMoves(b, 0).copy()
        if not v.any():
            v[:] = True
        pi = (rng.random(A) * v).astype(np.float32)
        pi /= pi.sum()
        with H.CounterRandom(seed=SYM_SEED, stream=j) as cr:
            sy = game.getSymmetries(b, pi, v)
            draws.append(cr.counter)
        assert len(sy) <= K
        for k, (s_, p_, v_) in enumerate(sy):
            out_s[j, k], out_p[j, k], out_v[j, k] = s_.reshape(-1), p_, np.asarray(v_).astype(np.uint8)
        states.append(b.reshape(-1)); pis.append(pi); vals.append(v.astype(np.uint8)); cnt.append(len(sy))
    return dict(state=np.array(states, dtype=np.int8), pi=np.array(pis, dtype=np.float32), valids=np.array(vals, dtype=np.uint8),
                count=np.array(cnt, dtype=np.int32), draws=np.array(draws, dtype=np.int32), out_state=out_s, out_pi=out_p,
                out_valids=out_v, seed=np.array(SYM_SEED), shape=env['shape'])


def gen_mcts(env, m, game, n):
    M = m['MCTS']
    H.enable_numba_typing(M)
    shape, A = tuple(env['shape']), n * n
    live = np.flatnonzero(~env['ended'].any(axis=1))
    roots = [env['init_boards'][0], env['canonical'][live[len(live) // 3]], env['canonical'][live[(5 * len(live)) // 6]]]
    cases = []
    for ri, root in enumerate(roots):
        for sims in (25, 300):
            for var in (dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True), dict(cpuct=1.25, fpu=0.1, universes=0, forced_playouts=False)):
                if ri == 1 and var['universes'] == 0:
                    continue
                args = H.mcts_args(m['utils'], numMCTSSims=sims, **var)
                mc = M.MCTS(game, H.HashNet(n), args)
                board = root.reshape(shape).copy()
                seed, stream = 5151, 100 + len(cases)
                with H.CounterRandom(seed=seed, stream=stream) as cr:
                    probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
                    draws = cr.counter
                nd = mc.nodes_data[board.tobytes()]
                cases.append(dict(root=root.copy(), sims=sims, cpuct=var['cpuct'], fpu=var['fpu'], universes=var['universes'],
                                  forced=int(var['forced_playouts']), Ns=nd[3], Qs=np.float32(nd[7]),
                                  Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                                  Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                                  q=np.asarray(q, dtype=np.float32), nodes=len(mc.nodes_data), digest=tree_digest(mc, A),
                                  rng_seed=seed, rng_stream=stream, rng_draws=draws))
                print('mcts case', len(cases) - 1, 'sims', sims, 'nodes', len(mc.nodes_data), 'draws', draws, flush=True)
    out = {'case_' + k: np.array([c[k] for c in cases]) for k in cases[0]}
    out['typed'] = np.array(1)
    return out


def main():
    for n in (3, 4, 5):
        env, m, game = gen_env(n, 6 if n == 3 else (4 if n == 4 else 3), seed=60 + n)
        np.savez_compressed(os.path.join(GOLDEN, 'env_tlp%d.npz' % n), **env)
        print('tlp', n, 'transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()),
              'refills', int((env['n_uniforms'] > 0).sum()))
        sym = gen_sym(env, game, n, 3 if n == 3 else (5 if n == 4 else 8))
        np.savez_compressed(os.path.join(GOLDEN, 'sym_tlp%d.npz' % n), **sym)
        print('  sym cases', len(sym['count']), 'forms', int(sym['count'].sum()), 'max', int(sym['count'].max()))
        if n == 3:
            np.savez_compressed(os.path.join(GOLDEN, 'mcts_tlp3_numba.npz'), **gen_mcts(env, m, game, n))
        H.cleanup()


if __name__ == '__main__':
    main()
