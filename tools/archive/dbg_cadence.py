import os, sys, numpy as np, torch
R = os.environ.get('AZG_ROOT', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from azg_amd import games
from azg_amd.selfplay import SelfPlayEngine
from hashnet import HashNetTorch
from tools_args import MCTS_ARGS
class Args(dict): __getattr__ = dict.get
g = games.SplendorGame(2)
args = Args(numMCTSSims=24, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=float(os.environ.get('ALPHA', '0.3')), temperature=[1.25, 0.8, 1.0], tempThreshold=6, **MCTS_ARGS['splendor2'])
T = 32
def run(K, graph, fused):
    e = SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=2048, max_examples=T * 400, rng_seed=99, stream0=7, use_graph=graph, advance_every=K, fused=fused)
    e.start()
    for _ in range(400):
        e.run(40); st = e.stats()
        assert st['errors'] == 0, st
        if st['games'] >= 3 * T: break
    ex = [x.cpu().numpy() for x in e.drain_examples()]
    meta = ex[5]; keep = np.flatnonzero(meta[:, 1] == 0); order = keep[np.lexsort((meta[keep, 2], meta[keep, 0]))]
    for grp in e.groups: grp.f.close()
    return [x[order] for x in ex]
cfgs = [(1, False, False), (5, False, True), (3, False, False), (5, True, True), (3, True, False), (1, True, True), (1, True, False)]
res = [run(*c) for c in cfgs]
for c, r in zip(cfgs[1:], res[1:]):
    same = all(np.array_equal(a, b) for a, b in zip(res[0], r)) if len(r[0]) == len(res[0][0]) else False
    msg = ''
    if not same and len(r[0]) == len(res[0][0]):
        for k in range(6):
            d = np.flatnonzero((res[0][k].reshape(len(r[0]), -1) != r[k].reshape(len(r[0]), -1)).any(axis=1))
            if len(d): msg += ' col%d first diff row %d meta %s n=%d;' % (k, d[0], res[0][5][d[0]], len(d))
    print(c, 'same' if same else 'DIFF', len(r[0]), msg)
