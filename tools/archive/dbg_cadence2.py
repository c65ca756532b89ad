"""debug: the self-play examples of 32 Splendor games under four advance cadences / launch forms must be identical (see
tests/test_gpu_selfplay.py test_advance_cadence_does_not_change_results); prints the per-run example counts."""
import os, sys, numpy as np, torch
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import importlib
from tools_args import MCTS_ARGS
class Args(dict): __getattr__=dict.get
sys.path.insert(0,'/root/repo')
import conftest  # noqa
from azg_amd import games
from azg_amd.selfplay import SelfPlayEngine
from hashnet import HashNetTorch
g = games.SplendorGame(2)
args = Args(numMCTSSims=24, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=float(os.environ.get('DBG_ALPHA', '0.3')), forced_playouts=bool(int(os.environ.get('DBG_FORCED', '1'))), temperature=[1.25, 0.8, 1.0], tempThreshold=6, **{k: v for k, v in MCTS_ARGS['splendor2'].items() if k != 'forced_playouts'})
T=32
import ctypes
from azg_amd._lib import lib
lib().azg_debug_poison_onchip.argtypes=[ctypes.c_uint32, ctypes.c_void_p]
PAT=[int(x,0) for x in sys.argv[2].split(',')] if len(sys.argv)>2 else None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    res=[]
    for K, graph, fused in ((1, False, False), (5, True, True), (3, True, False), (1, False, True)):
        if PAT: lib().azg_debug_poison_onchip(PAT[len(res) % len(PAT)], None)
        e = SelfPlayEngine(g, HashNetTorch(2), args, T, node_capacity=2048, max_examples=T * 400, rng_seed=99, stream0=7, use_graph=graph, advance_every=K, fused=fused)
        e.start()
        for _ in range(400):
            e.run(40)
            st = e.stats()
            assert st['errors'] == 0
            if st['games'] >= 3 * T: break
        ex = [x.cpu().numpy() for x in e.drain_examples()]
        meta = ex[5]
        keep = np.flatnonzero(meta[:, 1] == 0)
        order = keep[np.lexsort((meta[keep, 2], meta[keep, 0]))]
        res.append([x[order] for x in ex])
        for grp in e.groups: grp.f.close()
    lens=[len(r[0]) for r in res]
    # per-stream example counts
    per=[np.bincount(r[5][:,0]-7, minlength=T) for r in res]
    diff=[np.flatnonzero(per[0]!=p) for p in per]
    print('rep',rep,'lens',lens,'streams differing vs run0',[d.tolist() for d in diff])
