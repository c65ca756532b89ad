"""Golden vectors for the Akropolis plugin from the REFERENCE (imported live, pure-Python mode): the G1 / G3 / G5
families of tools/gen_golden.py (env transitions, MCTS traces with the hash-net under both operand typings + a tree-reuse sequence,
symmetries) for akropolis/AkropolisLogicNumba.py as shipped (N_PLAYERS = 2) and with N_PLAYERS = 3 / 4 (families akropolis3 / akropolis4).  Build-container only:
    python tools/gen_golden_akropolis.py

The tile refill draws with np.random.choice when random_seed == 0 (real moves and init_game, AkropolisLogicNumba.py:507-508); the
reference's global RNG is replaced by tools/refshim/harness.py CounterRandom fed with recorded uniforms (choice(a) = a[floor(u len)]),
so env_akropolis.npz has the same keys as the other env fixtures (`seed`, `uniforms`) plus `init_uniforms`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import gen_golden as G  # noqa: E402
import harness as H  # noqa: E402

G.VARIANTS['akropolis'] = (dict(), 'AkropolisGame', 'AkropolisGame')
G.MCTS_ARGS['akropolis'] = dict(cpuct=1.0, fpu=0.0, universes=1, forced_playouts=True)


def gen_env(n_traj, rng, n=2):
    m = H.load_reference(akropolis_players=n)
    with H.CounterRandom(injected=[0.5] * (n + 2)):
        game = m['AkropolisGame'].AkropolisGame()
    P, A = n, game.getActionSize()
    keys = ('state', 'player', 'valid', 'action', 'seed', 'next_state', 'next_player', 'ended', 'score', 'round', 'canonical', 'uniforms',
            'n_uniforms', 'traj')
    rec = {k: [] for k in keys}
    inits, init_u = [], []
    for t in range(n_traj):
        us = [float(x) for x in rng.random(n + 2)]
        with H.CounterRandom(injected=list(us)) as cr:
            board = game.getInitBoard().copy()
            assert len(cr.used) == n + 2
        inits.append(board.copy()); init_u.append(us)
        player = 0
        for ply in range(100):
            valid = game.getValidMoves(board, player).copy()
            idx = np.flatnonzero(valid)
            # stack on existing tiles and pay stones for later slots fairly often (heights, quarries, plazas under tiles)
            r = rng.random()
            heights = board[:, :, n + player].reshape(-1)
            on_top = idx[heights[(idx % 1014) // 6] > 0]
            a = int(rng.choice(on_top)) if (len(on_top) and r < 0.35) else int(rng.choice(idx[idx >= 1014])) if (r < 0.6 and (idx >= 1014).any()) \
                else int(rng.choice(idx))
            r = rng.random()
            seed = 0 if r < 0.3 else (-1 if r < 0.4 else H.MAGIC_SEEDS[int(rng.integers(8))])
            us = [float(x) for x in rng.random(n + 2)]
            with H.CounterRandom(injected=list(us)) as cr:
                nb, npl = game.getNextState(board, player, a, random_seed=seed)
                used = list(cr.used)
            nb = nb.copy()
            ended = game.getGameEnded(nb, npl).copy()
            rec['state'].append(board.reshape(-1).copy()); rec['player'].append(player)
            rec['valid'].append(np.packbits(valid.astype(np.uint8))); rec['action'].append(a); rec['seed'].append(seed)
            rec['next_state'].append(nb.reshape(-1).copy()); rec['next_player'].append(npl)
            rec['ended'].append(ended.astype(np.float32))
            rec['score'].append([int(game.getScore(nb, p)) for p in range(P)])
            rec['round'].append(int(game.getRound(nb)))
            rec['canonical'].append(game.getCanonicalForm(nb, npl).reshape(-1).copy())
            rec['uniforms'].append((used + [0.5] * (n + 2))[:n + 2]); rec['n_uniforms'].append(len(used))
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
        shape=np.array(game.getBoardSize()), A=np.array(A), P=np.array(P))
    return out, m, game


def main():
    for n in (2, 3, 4):
        tag = 'akropolis' if n == 2 else 'akropolis%d' % n
        G.VARIANTS[tag] = (dict(akropolis_players=n), 'AkropolisGame', 'AkropolisGame')
        G.MCTS_ARGS[tag] = G.MCTS_ARGS['akropolis']
        rng = np.random.default_rng(sum(map(ord, tag)))
        env, m, game = gen_env(6 if n == 2 else 4, rng, n)
        np.savez_compressed(os.path.join(G.GOLDEN, 'env_%s.npz' % tag), **env)
        st = env['next_state'].reshape((-1, 13, 13, 3 * n + 2))
        print(tag, 'env transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()), 'max score', int(env['score'].max()),
              'max height', int(st[:, :, :, n:2 * n].max()), 'seed-0 refills', int(((env['seed'] == 0) & (env['n_uniforms'] > 0)).sum()),
              'districts seen', sorted(set(np.flatnonzero(st[:, n:2 * n, :5, 3 * n].reshape(-1, 5).max(axis=0) > 0).tolist())))
        sym = G.gen_sym(tag, env, game, rng, 3)
        np.savez_compressed(os.path.join(G.GOLDEN, 'sym_%s.npz' % tag), **sym)
        for typed in ((0, 1) if n == 2 else (1,)):
            mc = G.gen_mcts(tag, env, m, game, rng, sims_list=[25, 200], n_roots=2, seq_moves=12, typed=typed)
            np.savez_compressed(os.path.join(G.GOLDEN, 'mcts_%s_%s.npz' % (tag, 'numba' if typed else 'numpy2')), **mc)
            print(tag, 'mcts cases', len(mc['case_sims']), 'seq', len(mc['seq_action']), 'typed', typed)
        H.cleanup()


if __name__ == '__main__':
    main()
