"""Headline-size MCTS golden vectors from the REFERENCE (imported live, pure-Python mode, Numba operand typing):
tests/golden/mcts800_<variant>_numba.npz -- numMCTSSims = 800 (BASELINE.json's search size) from two roots per variant
(the first captured init board and one mid-game canonical state of env_<variant>.npz), the checkpoint's MCTS args, the
integer hash-net of SURVEY.md Appendix C.3.  Same schema as the case_* arrays of mcts_<variant>_numba.npz
(tools/gen_golden.py); kept in its own files so that the round-1 fixtures stay byte-identical.

Build-container only (needs /root/reference):  python tools/gen_golden_800.py [--only VARIANT]

BASELINE config 5 (Azul, numMCTSSims = 1600, Dirichlet noise with the automatic alpha):
    python tools/gen_golden_800.py --only azul --sims 1600 --noise      ->  tests/golden/mcts1600_azul_numba.npz
adds `noise_*` arrays: the same two roots searched by an MCTS built with dirichlet_noise=True and args.dirichletAlpha = -1
(alpha = 10 / n_valid, MCTS.py:188-192); the sample `rng.dirichlet([alpha] * n_valid)` drew is recorded (the generator is a seeded
numpy Generator behind a recording wrapper), so that oracle and engine can be handed the same sample.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
import harness as H  # noqa: E402
from gen_golden import GOLDEN, MCTS_ARGS, VARIANTS, tree_digest  # noqa: E402

SIMS = 800


class RecordingRng:
    """numpy Generator whose dirichlet() calls are recorded (alpha vector, sample)"""

    def __init__(self, seed):
        self.g, self.calls = np.random.default_rng(seed), []

    def dirichlet(self, alpha):
        s = self.g.dirichlet(alpha)
        self.calls.append((np.asarray(alpha, dtype=np.float64).copy(), np.asarray(s, dtype=np.float64).copy()))
        return s

    def random(self):
        return self.g.random()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--sims', type=int, default=SIMS)
    ap.add_argument('--noise', action='store_true')
    a = ap.parse_args()
    sims = a.sims
    for name in ('splendor2', 'splendor4', 'santorini1', 'santorini11', 'azul'):
        if a.only and name != a.only:
            continue
        kw, modkey, cls = VARIANTS[name]
        m = H.load_reference(**kw)
        game = getattr(m[modkey], cls)()
        M = m['MCTS']
        H.enable_numba_typing(M)
        env = np.load(os.path.join(GOLDEN, 'env_%s.npz' % name))
        shape = tuple(env['shape'])
        live = np.flatnonzero(~env['ended'].any(axis=1))
        mid = int(live[len(live) // 2])
        roots = [env['init_boards'][0].copy(), env['canonical'][mid].copy()]
        cases = []
        for root in roots:
            margs = dict(MCTS_ARGS[name])
            args = H.mcts_args(m['utils'], numMCTSSims=sims, **margs)
            mc = M.MCTS(game, H.HashNet(game.num_players), args)
            board = root.reshape(shape)
            probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
            nd = mc.nodes_data[board.tobytes()]
            cases.append(dict(root=root, sims=sims, cpuct=margs['cpuct'], fpu=margs['fpu'], universes=margs['universes'],
                              forced=int(margs['forced_playouts']), Ns=nd[3], Qs=np.float32(nd[7]),
                              Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                              Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                              q=np.asarray(q, dtype=np.float32), nodes=len(mc.nodes_data),
                              digest=tree_digest(mc, game.getActionSize())))
            print(name, 'root', len(cases) - 1, 'Ns', nd[3], 'nodes', len(mc.nodes_data), flush=True)
        out = {'case_' + k: np.array([c[k] for c in cases]) for k in cases[0]}
        out['typed'] = np.array(1)
        if a.noise:
            A = game.getActionSize()
            nz = []
            for ri, root in enumerate(roots):
                margs = dict(MCTS_ARGS[name])
                args = H.mcts_args(m['utils'], numMCTSSims=sims, dirichletAlpha=-1, temperature=[1.0, 1.0, 1.0], **margs)
                mc = M.MCTS(game, H.HashNet(game.num_players), args, dirichlet_noise=True)
                mc.rng = RecordingRng(1600 + ri)
                board = root.reshape(shape)
                probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
                nd = mc.nodes_data[board.tobytes()]
                assert len(mc.rng.calls) == 1
                alpha, sample = mc.rng.calls[0]
                nv = int(np.asarray(game.getValidMoves(board, 0)).sum())
                assert len(alpha) == nv and np.all(alpha == 10.0 / nv)                       # the automatic value, MCTS.py:190-192
                pad = np.zeros(A, dtype=np.float64)
                pad[:nv] = sample
                nz.append(dict(root=root, alpha=alpha[0], n_valid=nv, sample=pad, Ns=nd[3], Qs=np.float32(nd[7]),
                               Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                               Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                               nodes=len(mc.nodes_data), digest=tree_digest(mc, A)))
                print(name, 'noise root', ri, 'alpha', alpha[0], 'Ns', nd[3], flush=True)
            out.update({'noise_' + k: np.array([c[k] for c in nz]) for k in nz[0]})
        np.savez_compressed(os.path.join(GOLDEN, 'mcts%d_%s_numba.npz' % (sims, name)), **out)
        H.cleanup()


if __name__ == '__main__':
    main()
