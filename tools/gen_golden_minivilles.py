"""Golden vectors for the Minivilles plugin from the REFERENCE (imported live, pure-Python mode).
Build-container only:  python tools/gen_golden_minivilles.py

Minivilles draws true randomness inside make_move whatever random_seed says (dice: np.random.randint, purple cards:
np.random.random -- MinivillesLogicNumba.py:49-52,232-242), in MCTS simulations too.  The reference's global RNG is replaced by
the counter-based stream of the engine's RNG contract (tools/refshim/harness.py CounterRandom), so that every uniform the
reference consumes is known: what is pinned is how each uniform is consumed, and in which order.

  env_minivilles<n>.npz       seeded-play transitions (state, player, valid mask, action, next state, next player, ended, score,
                              round, canonical form) + the uniforms each step consumed + captured init boards with theirs
  mcts_minivilles2_numba.npz  MCTS.getActionProb traces (hash-net, Numba operand typing) with the search-time dice drawn from a
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

MAXU = 8          # a step consumes at most 2 (dice) + 3 (business centre) + 1 (TV) uniforms


def gen_env(n_players, n_traj, seed):
    m = H.load_reference(minivilles_players=n_players)
    rec = {k: [] for k in ('state', 'player', 'valid', 'action', 'next_state', 'next_player', 'ended', 'score', 'round', 'canonical',
                           'uniforms', 'n_uniforms', 'traj')}
    inits, init_u = [], []
    rng = np.random.default_rng(seed)
    with H.CounterRandom(seed=777, stream=0):
        game = m['MinivillesGame'].MinivillesGame()          # the constructor rolls dice too
    P = game.num_players
    for t in range(n_traj):
        with H.CounterRandom(seed=1000 + n_players, stream=t) as cr:
            board = game.getInitBoard().copy()
            iu = list(cr.used)
        inits.append(board.copy()); init_u.append((iu + [0.5] * MAXU)[:MAXU])
        player = 0
        ctr = len(iu)
        for ply in range(400):
            valid = game.getValidMoves(board, player).copy()
            idx = np.flatnonzero(valid)
            buy = idx[idx < 19]
            # mostly buy (the dearest affordable thing fairly often) so that monuments, purple cards and the end are reached
            if len(buy) and rng.random() < 0.8:
                a = int(buy[-1]) if rng.random() < 0.5 else int(rng.choice(buy))
            else:
                a = int(rng.choice(idx))
            with H.CounterRandom(seed=1000 + n_players, stream=t, counter=ctr) as cr:
                nb, npl = game.getNextState(board, player, a, random_seed=int(rng.integers(0, 3)))
                used = list(cr.used)
                ctr = cr.counter
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy()); rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8))); rec['action'].append(a)
            rec['next_state'].append(nb.reshape(-1).copy()); rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(P)])
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
               init_uniforms=np.array(init_u, dtype=np.float64), shape=np.array(game.getBoardSize()), A=np.array(21), P=np.array(P))
    return out, m, game


def gen_mcts(env, m, game):
    M = m['MCTS']
    H.enable_numba_typing(M)
    shape = tuple(env['shape'])
    live = np.flatnonzero(~env['ended'].any(axis=1))
    roots = [env['init_boards'][0], env['canonical'][live[len(live) // 3]], env['canonical'][live[2 * len(live) // 3]]]
    cases = []
    for ri, root in enumerate(roots):
        for sims in (25, 200):
            for var in (dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True), dict(cpuct=1.25, fpu=0.1, universes=0, forced_playouts=False)):
                if ri > 0 and var['universes'] == 0:
                    continue
                args = H.mcts_args(m['utils'], numMCTSSims=sims, **var)
                mc = M.MCTS(game, H.HashNet(game.num_players), args)
                board = root.reshape(shape).copy()
                seed, stream = 4242, 100 + len(cases)
                with H.CounterRandom(seed=seed, stream=stream) as cr:
                    probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
                    draws = cr.counter
                nd = mc.nodes_data[board.tobytes()]
                cases.append(dict(root=root.copy(), sims=sims, cpuct=var['cpuct'], fpu=var['fpu'], universes=var['universes'],
                                  forced=int(var['forced_playouts']), Ns=nd[3], Qs=np.float32(nd[7]),
                                  Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                                  Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                                  q=np.asarray(q, dtype=np.float32), nodes=len(mc.nodes_data), digest=tree_digest(mc, 21),
                                  rng_seed=seed, rng_stream=stream, rng_draws=draws))
                print('mcts case', len(cases) - 1, 'sims', sims, 'nodes', len(mc.nodes_data), 'draws', draws, flush=True)
    out = {'case_' + k: np.array([c[k] for c in cases]) for k in cases[0]}
    out['typed'] = np.array(1)
    return out


def main():
    for n in (2, 3, 4):
        env, m, game = gen_env(n, 8 if n == 2 else 3, seed=50 + n)
        np.savez_compressed(os.path.join(GOLDEN, 'env_minivilles%d.npz' % n), **env)
        print('minivilles', n, 'transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()),
              'max uniforms/step', int(env['n_uniforms'].max()))
        if n == 2:
            mc = gen_mcts(env, m, game)
            np.savez_compressed(os.path.join(GOLDEN, 'mcts_minivilles2_numba.npz'), **mc)
        H.cleanup()


if __name__ == '__main__':
    main()
