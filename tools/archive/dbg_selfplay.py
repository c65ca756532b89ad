import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
from azg_amd import games
from azg_amd.forest import Forest
from hashnet import HashNetTorch
class Args(dict): __getattr__ = dict.get
g = games.SplendorGame(2)
args = Args(numMCTSSims=40, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1.25,0.8,1.0], tempThreshold=6, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True)
T=16
f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=int(sys.argv[1]) if len(sys.argv)>1 else 2048, max_examples=T*300, rng_seed=4242, stream0=1000)
net = HashNetTorch(2)
f.selfplay_start(); torch.cuda.synchronize(); print('start ok', flush=True)
for rnd in range(20000):
    dbg = rnd >= int(os.environ.get('DBG_FROM', '1000000'))
    if dbg: print(rnd, 'select', flush=True)
    f.select(); torch.cuda.synchronize()
    pi, vv = net.predict_batch(f.leaf_states.view((T,)+f.board_shape()), f.leaf_valid.bool())
    if dbg: print(rnd, 'expand', flush=True)
    f.expand_backup(pi, vv); torch.cuda.synchronize()
    if dbg: print(rnd, 'advance', f.stats()['max_nodes'], flush=True)
    f.selfplay_advance(); torch.cuda.synchronize()
    if dbg:
        nb = f.validate()
        print(rnd, 'validate', nb, f.stats()['gc_runs'], flush=True)
        if nb: break
    if rnd % 500 == 0:
        print(rnd, f.stats(), flush=True)
print('done', f.stats())
