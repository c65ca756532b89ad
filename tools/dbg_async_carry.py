import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from azg_amd import games
from azg_amd.selfplay import SelfPlayEngine
from hashnet import HashNetPipeline
class Args(dict):
    __getattr__ = dict.get
g = games.SplendorGame(2)
T, sims = int(os.environ.get('TT', 1024)), 24
args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
            cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=False)
runs = []
for stall in (os.environ.get('STALL', '5'), None):
    if stall: os.environ['AZG_ASYNC_TEST_STALL'] = stall
    else: os.environ.pop('AZG_ASYNC_TEST_STALL', None)
    e = SelfPlayEngine(g, HashNetPipeline(2), args, T, node_capacity=2048, max_examples=T * 200, rng_seed=78, stream0=1200,
                       async_pipe=True, deterministic=True, async_cfg=dict(n_net=16, n_sel=int(os.environ.get('NSEL', 32))))
    e.start(episode_quota=T)
    trace = []
    for k in range(30):
        e.run(61)
        st = e.stats()
        assert st['errors'] == 0, st
        trace.append((st['games'], st['plies'], st['sims']))
    to = e.forest.async_profile(reset=False)['timeouts']
    runs.append((trace, to))
    e.close()
print('timeouts', runs[0][1], runs[1][1])
bad = [(i, a, b) for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])) if a != b]
print('first mismatches', bad[:5], 'of', len(bad))
