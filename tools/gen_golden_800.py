"""Headline-size MCTS golden vectors from the REFERENCE (imported live, pure-Python mode, Numba operand typing):
tests/golden/mcts800_<variant>_numba.npz -- numMCTSSims = 800 (BASELINE.json's search size) from two roots per variant
(the first captured init board and one mid-game canonical state of env_<variant>.npz), the checkpoint's MCTS args, the
integer hash-net of SURVEY.md Appendix C.3.  Same schema as the case_* arrays of mcts_<variant>_numba.npz
(tools/gen_golden.py); kept in its own files so that the round-1 fixtures stay byte-identical.

Build-container only (needs /root/reference):  python tools/gen_golden_800.py [--only VARIANT]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
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
            args = H.mcts_args(m['utils'], numMCTSSims=SIMS, **margs)
            mc = M.MCTS(game, H.HashNet(game.num_players), args)
            board = root.reshape(shape)
            probs, q, full = mc.getActionProb(board, temp=1, force_full_search=True)
            nd = mc.nodes_data[board.tobytes()]
            cases.append(dict(root=root, sims=SIMS, cpuct=margs['cpuct'], fpu=margs['fpu'], universes=margs['universes'],
                              forced=int(margs['forced_playouts']), Ns=nd[3], Qs=np.float32(nd[7]),
                              Nsa=np.asarray(nd[5], dtype=np.int64), Qsa=np.asarray(nd[4], dtype=np.float64),
                              Ps=np.asarray(nd[2], dtype=np.float32), probs=np.asarray(probs, dtype=np.float64),
                              q=np.asarray(q, dtype=np.float32), nodes=len(mc.nodes_data),
                              digest=tree_digest(mc, game.getActionSize())))
            print(name, 'root', len(cases) - 1, 'Ns', nd[3], 'nodes', len(mc.nodes_data), flush=True)
        out = {'case_' + k: np.array([c[k] for c in cases]) for k in cases[0]}
        out['typed'] = np.array(1)
        np.savez_compressed(os.path.join(GOLDEN, 'mcts800_%s_numba.npz' % name), **out)
        H.cleanup()


if __name__ == '__main__':
    main()
