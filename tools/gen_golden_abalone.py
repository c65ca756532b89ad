"""Golden vectors for the Abalone plugin from the REFERENCE (imported live, pure-Python mode): the G1 / G3 / G5
families of tools/gen_golden.py (env transitions, MCTS traces with the hash-net under both operand typings, symmetries) for
abalone/AbaloneLogicNumba.py as shipped (INITIAL_LAYOUT = 1, no dynamic komi).  Build-container only:
    python tools/gen_golden_abalone.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import gen_golden as G  # noqa: E402
import harness as H  # noqa: E402

G.VARIANTS['abalone'] = (dict(), 'AbaloneGame', 'AbaloneGame')
G.MCTS_ARGS['abalone'] = dict(cpuct=1.0, fpu=0.0, universes=0, forced_playouts=True)


def main():
    rng = np.random.default_rng(sum(map(ord, 'abalone')))
    env, m, game = G.gen_env('abalone', 5, rng, max_plies=140)
    np.savez_compressed(os.path.join(G.GOLDEN, 'env_abalone.npz'), **env)
    print('abalone env transitions', len(env['state']), 'ended', int(env['ended'].any(axis=1).sum()), 'max score',
          int(env['score'].max()))
    sym = G.gen_sym('abalone', env, game, rng, 3)
    np.savez_compressed(os.path.join(G.GOLDEN, 'sym_abalone.npz'), **sym)
    for typed in (0, 1):
        mc = G.gen_mcts('abalone', env, m, game, rng, sims_list=[25, 200], n_roots=2, seq_moves=12, typed=typed)
        np.savez_compressed(os.path.join(G.GOLDEN, 'mcts_abalone_%s.npz' % ('numba' if typed else 'numpy2')), **mc)
        print('abalone mcts cases', len(mc['case_sims']), 'seq', len(mc['seq_action']), 'typed', typed)
    H.cleanup()


if __name__ == '__main__':
    main()
