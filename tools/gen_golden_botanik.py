"""Golden vectors for the Botanik plugin from the REFERENCE (imported live, pure-Python mode).
Build-container only:  python tools/gen_golden_botanik.py

Every card drawn from the deck inside make_move / init_game (BotanikLogicNumba.py:414-438) takes one uniform of the reference's
global RNG, in MCTS simulations too; it is replaced by the counter stream of the engine's RNG contract (tools/refshim/harness.py
CounterRandom) so that every uniform consumed is known.

  env_botanik.npz          transitions of mixed greedy / random play (state, player, valid mask, action, next state, next player,
                           ended, score, round, canonical form) + the uniforms each step consumed + captured init boards with theirs
  sym_botanik.npz          get_symmetries of canonical states (11 to 14 forms)
  mcts_botanik_numba.npz   MCTS.getActionProb traces (hash-net, Numba operand typing), the draws inside the search from a recorded
                           stream: root statistics, probs, q, node count, whole-tree digest, draws consumed
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402
from gen_golden import GOLDEN, tree_digest  # noqa: E402

MAXU = 8          # init draws 8 cards; a step draws 0 or 3
A = 428


def pick_move(game, board, player, valid, rng):
    """mixed greedy / random play that fills the registers, unlinks cards through the middle row and grows scoring machines, so
    that every status, the mecabot swaps, area merges and the end-of-game branches occur"""
    idx = np.flatnonzero(valid)
    if len(idx) == 1 or rng.random() < 0.2:
        return int(rng.choice(idx))
    best, best_a = -1e9, int(idx[0])
    for a in idx:
        with H.CounterRandom(seed=1, stream=0):
            nb, _ = game.getNextState(board, player, int(a))
        me, op = player, 1 - player
        mach = nb[6 + 10 * me:16 + 10 * me].reshape(-1)[:343].reshape(7, 7, 7)
        val = (2.0 * (float(nb[0, 1, me]) - float(nb[0, 1, op])) + 0.6 * np.count_nonzero(mach[:, :, 0])
               + 0.5 * np.count_nonzero(nb[5, 2 * me:2 * me + 2, 0]) + 0.25 * np.count_nonzero(nb[2 + me, :, 0]) + rng.random())
        if val > best:
            best, best_a = val, int(a)
    return best_a


def gen_env(n_traj, seed):
    m = H.load_reference()
    keys = ('state', 'player', 'valid', 'action', 'next_state', 'next_player', 'ended', 'score', 'round', 'canonical', 'uniforms',
            'n_uniforms', 'traj')
    rec = {k: [] for k in keys}
    inits, init_u = [], []
    rng = np.random.default_rng(seed)
    with H.CounterRandom(seed=777, stream=0):
        game = m['BotanikGame'].BotanikGame()
    for t in range(n_traj):
        with H.CounterRandom(seed=3000, stream=t) as cr:
            board = game.getInitBoard().copy()
            iu = list(cr.used)
        inits.append(board.copy()); init_u.append((iu + [0.5] * MAXU)[:MAXU])
        player, ctr = 0, len(iu)
        for ply in range(400):
            valid = game.getValidMoves(board, player).copy()
            a = pick_move(game, board, player, valid, rng)
            with H.CounterRandom(seed=3000, stream=t, counter=ctr) as cr:
                nb, npl = game.getNextState(board, player, a, random_seed=int(rng.integers(0, 3)))
                used, ctr = list(cr.used), cr.counter
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy()); rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8))); rec['action'].append(a)
            rec['next_state'].append(nb.reshape(-1).copy()); rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(2)])
            rec['round'].append(int(game.getRound(nb)))
            rec['canonical'].append(game.getCanonicalForm(nb, npl).reshape(-1).copy())
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
               init_uniforms=np.array(init_u, dtype=np.float64), shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(2))
    return out, m, game


def gen_sym(env, game):
    """get_symmetries on canonical states: every state where both freed cards of player 0 are present (the 14-form case) and
    every 6th other one; pi = a random policy (not masked: the maps must move every entry)"""
    shape = tuple(env['shape'])
    rng = np.random.default_rng(78)
    can = env['canonical'].reshape((-1,) + shape)
    both = np.flatnonzero((can[:, 5, 0, 0] != 0) & (can[:, 5, 1, 0] != 0))
    rows = np.unique(np.concatenate([both[:12], np.arange(0, len(can), 25)]))
    K = 14
    out_s = np.zeros((len(rows), K, can[0].size), dtype=np.int8)
    out_p = np.zeros((len(rows), K, A), dtype=np.float32)
    out_v = np.zeros((len(rows), K, A), dtype=np.uint8)
    states, pis, vals, cnt = [], [], [], []
    for j, i in enumerate(rows):
        b = can[i].copy()
        v = game.getValidMoves(b, 0).copy()
        pi = rng.random(A).astype(np.float32)
        pi /= pi.sum()
        sy = game.getSymmetries(b, pi, v.copy())
        assert len(sy) <= K
        for k, (s_, p_, v_) in enumerate(sy):
            out_s[j, k], out_p[j, k], out_v[j, k] = s_.reshape(-1), p_, np.asarray(v_).astype(np.uint8)
        states.append(b.reshape(-1)); pis.append(pi); vals.append(v.astype(np.uint8)); cnt.append(len(sy))
    return dict(state=np.array(states, dtype=np.int8), pi=np.array(pis, dtype=np.float32), valids=np.array(vals, dtype=np.uint8),
                count=np.array(cnt, dtype=np.int32), out_state=out_s, out_pi=out_p, out_valids=out_v, shape=env['shape'])


def gen_mcts(env, m, game):
    M = m['MCTS']
    H.enable_numba_typing(M)
    shape = tuple(env['shape'])
    live = np.flatnonzero(~env['ended'].any(axis=1))
    status = env['canonical'].reshape((-1,) + shape)[:, 0, 0, 1]
    expand = [i for i in live if status[i] in (1, 3)]
    roots = [env['init_boards'][0], env['canonical'][live[len(live) // 3]], env['canonical'][expand[len(expand) // 2]],
             env['canonical'][live[(9 * len(live)) // 10]]]
    cases = []
    for ri, root in enumerate(roots):
        for sims in (25, 300):
            for var in (dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True), dict(cpuct=1.25, fpu=0.1, universes=0, forced_playouts=False)):
                if ri in (1, 3) and var['universes'] == 0:
                    continue
                args = H.mcts_args(m['utils'], numMCTSSims=sims, **var)
                mc = M.MCTS(game, H.HashNet(2), args)
                board = root.reshape(shape).copy()
                seed, stream = 6161, 100 + len(cases)
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
    env, m, game = gen_env(8, seed=90)
    np.savez_compressed(os.path.join(GOLDEN, 'env_botanik.npz'), **env)
    st = env['next_state'].reshape((-1,) + tuple(env['shape']))
    print('botanik transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()), 'draw steps', int((env['n_uniforms'] > 0).sum()),
          'max score', int(env['score'].max()), 'statuses', sorted(set(st[:, 0, 0, 1].tolist())), 'throw-away moves', int((env['action'] == 427).sum()),
          'results', sorted(set(map(tuple, env['ended'][env['ended'].any(axis=1)].tolist()))))
    sym = gen_sym(env, game)
    np.savez_compressed(os.path.join(GOLDEN, 'sym_botanik.npz'), **sym)
    print('  sym cases', len(sym['count']), 'forms', int(sym['count'].sum()), 'with 14 forms', int((sym['count'] == 14).sum()))
    np.savez_compressed(os.path.join(GOLDEN, 'mcts_botanik_numba.npz'), **gen_mcts(env, m, game))
    H.cleanup()


if __name__ == '__main__':
    main()
