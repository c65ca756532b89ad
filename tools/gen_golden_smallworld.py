"""Golden vectors for the Smallworld plugin from the REFERENCE (imported live, pure-Python mode): the G1 / G3 / G5
families of tools/gen_golden.py (env transitions, MCTS traces with the hash-net under both operand typings + a tree-reuse sequence,
symmetries) for smallworld/SmallworldLogicNumba.py as shipped (NUMBER_PLAYERS = 2).  Build-container only:
    python tools/gen_golden_smallworld.py

Dice and deck draws use np.random.choice when random_seed == 0 (real moves and init_game) and get_symmetries draws two score offsets
with np.random.randint; the reference's global RNG is replaced by tools/refshim/harness.py CounterRandom (choice(a) = a[floor(u len)],
randint(lo, hi) = lo + floor(u (hi - lo))): env steps are fed recorded uniforms (`seed`, `uniforms`, `init_uniforms`), the symmetries of
case j draw from the counter stream (seed, j)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import gen_golden as G  # noqa: E402
import harness as H  # noqa: E402

G.VARIANTS['smallworld'] = (dict(), 'SmallworldGame', 'SmallworldGame')
G.MCTS_ARGS['smallworld'] = dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True)
NU, NU_INIT, SYM_SEED = 4, 12, 808


def gen_env(n_traj, rng, n=2):
    m = H.load_reference(smallworld_players=n)
    with H.CounterRandom(injected=[0.5] * NU_INIT):
        game = m['SmallworldGame'].SmallworldGame()
    A = game.getActionSize()
    keys = ('state', 'player', 'valid', 'action', 'seed', 'next_state', 'next_player', 'ended', 'score', 'round', 'canonical', 'uniforms',
            'n_uniforms', 'traj')
    rec = {k: [] for k in keys}
    inits, init_u = [], []
    for t in range(n_traj):
        us = [float(x) for x in rng.random(NU_INIT)]
        with H.CounterRandom(injected=list(us)) as cr:
            board = game.getInitBoard().copy()
            assert len(cr.used) == NU_INIT
        inits.append(board.copy()); init_u.append(us)
        player = 0
        for ply in range(3000):
            valid = game.getValidMoves(board, player).copy()
            idx = np.flatnonzero(valid)
            # conquests / people and power actions three times as often, rarely end or decline early
            w = np.ones(len(idx)); w[idx == A - 1] = 0.15; w[idx == A - 2] = 0.1; w[(idx >= (A - 16) // 5) & (idx < 4 * ((A - 16) // 5))] = 3.0
            a = int(rng.choice(idx, p=w / w.sum()))
            r = rng.random()
            seed = 0 if r < 0.3 else (-1 if r < 0.4 else H.MAGIC_SEEDS[int(rng.integers(8))])
            us = [float(x) for x in rng.random(NU)]
            with H.CounterRandom(injected=list(us)) as cr:
                nb, npl = game.getNextState(board, player, a, random_seed=seed)
                used = list(cr.used)
            assert len(used) <= NU
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy()); rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8))); rec['action'].append(a); rec['seed'].append(seed)
            rec['next_state'].append(nb.reshape(-1).copy()); rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(n)])
            rec['round'].append(int(game.getRound(nb)))
            rec['canonical'].append(game.getCanonicalForm(nb, npl).reshape(-1).copy())
            rec['uniforms'].append((used + [0.5] * NU)[:NU]); rec['n_uniforms'].append(len(used))
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
        n_uniforms=np.array(rec['n_uniforms'], dtype=np.int8), traj=np.array(rec['traj'], dtype=np.int16),
        init_boards=np.array([b.reshape(-1) for b in inits], dtype=np.int8), init_uniforms=np.array(init_u, dtype=np.float64),
        shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(n))
    return out, m, game


def gen_sym(env, game, every):
    A = int(env['A'])
    rng = np.random.default_rng(79)
    shape = tuple(env['shape'])
    rows = np.arange(0, len(env['canonical']), every)
    S = env['state'].shape[1]
    out_s = np.zeros((len(rows), 3, S), dtype=np.int8)
    out_p = np.zeros((len(rows), 3, A), dtype=np.float32)
    out_v = np.zeros((len(rows), 3, A), dtype=np.uint8)
    states, pis, vals, cnt, draws = [], [], [], [], []
    for j, i in enumerate(rows):
        b = env['canonical'][i].reshape(shape).copy()
        v = game.getValidMoves(b, 0).copy()
        pi = rng.random(A).astype(np.float32)
        pi /= pi.sum()
        with H.CounterRandom(seed=SYM_SEED, stream=j) as cr:
            sy = game.getSymmetries(b, pi, v.copy())
            draws.append(cr.counter)
        for k, (s_, p_, v_) in enumerate(sy):
            out_s[j, k], out_p[j, k], out_v[j, k] = s_.reshape(-1), p_, np.asarray(v_).astype(np.uint8)
        states.append(b.reshape(-1)); pis.append(pi); vals.append(v.astype(np.uint8)); cnt.append(len(sy))
    return dict(state=np.array(states, dtype=np.int8), pi=np.array(pis, dtype=np.float32), valids=np.array(vals, dtype=np.uint8),
                count=np.array(cnt, dtype=np.int32), draws=np.array(draws, dtype=np.int32), out_state=out_s, out_pi=out_p, out_valids=out_v,
                seed=np.array(SYM_SEED), shape=env['shape'])


def main():
    for n in (2, 3, 4):
        tag = 'smallworld' if n == 2 else 'smallworld%d' % n
        G.VARIANTS[tag] = (dict(smallworld_players=n), 'SmallworldGame', 'SmallworldGame')
        G.MCTS_ARGS[tag] = G.MCTS_ARGS['smallworld']
        rng = np.random.default_rng(sum(map(ord, tag)))
        env, m, game = gen_env(24 if n == 2 else 6, rng, n)
        np.savez_compressed(os.path.join(G.GOLDEN, 'env_%s.npz' % tag), **env)
        na = (int(env['A']) - 16) // 5
        st = env['next_state'].reshape((-1,) + tuple(env['shape']))
        print(tag, 'env transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()), 'max score', int(env['score'].max()),
              'peoples', len(set(np.abs(st[:, na:na + 3 * n, 1]).reshape(-1).tolist())) - 1, 'powers',
              len(set(np.abs(st[:, na:na + 3 * n, 2]).reshape(-1).tolist())) - 1, 'seed-0 draws',
              int(((env['seed'] == 0) & (env['n_uniforms'] > 0)).sum()), 'max uniforms', int(env['n_uniforms'].max()))
        sym = gen_sym(env, game, 40)
        np.savez_compressed(os.path.join(G.GOLDEN, 'sym_%s.npz' % tag), **sym)
        print('  sym cases', len(sym['count']), 'forms', int(sym['count'].sum()))
        for typed in ((0, 1) if n == 2 else (1,)):
            mc = G.gen_mcts(tag, env, m, game, rng, sims_list=[25, 200], n_roots=2, seq_moves=12, typed=typed)
            np.savez_compressed(os.path.join(G.GOLDEN, 'mcts_%s_%s.npz' % (tag, 'numba' if typed else 'numpy2')), **mc)
            print(tag, 'mcts cases', len(mc['case_sims']), 'seq', len(mc['seq_action']), 'typed', typed)
        H.cleanup()


if __name__ == '__main__':
    main()
