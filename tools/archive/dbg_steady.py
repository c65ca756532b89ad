"""Per-kernel time of one round in the desynchronised steady state (after all trees have restarted at least once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from azg_amd import games
from azg_amd.nnet import SplendorV80Hip
from azg_amd.selfplay import SelfPlayEngine
class Args(dict): __getattr__ = dict.get
a = Args(numMCTSSims=800, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0],
         tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
T = 4096
g = games.SplendorGame(2)
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=T)
e = SelfPlayEngine(g, net, a, T, node_capacity=int(os.environ.get('CAP', '13312')), max_examples=T * 160, work_budget=int(os.environ.get('WB', '0')))
e.start(); e.run(int(sys.argv[1]) if len(sys.argv) > 1 else 56000)
torch.cuda.synchronize()
grp = e.groups[0]; f = grp.f
def timed(fn):
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); t.record(); t.synchronize(); return s.elapsed_time(t) * 1000
acc = dict(select=[], net_expand=[], advance=[])
for r in range(64):
    acc['select'].append(timed(grp.select))
    acc['net_expand'].append(timed(grp.predict_expand))
    acc['advance'].append(timed(f.selfplay_advance))
import numpy as np
for k, v in acc.items():
    v = np.array(v); print('%-11s mean %8.1f us  p50 %8.1f  max %8.1f' % (k, v.mean(), np.percentile(v, 50), v.max()))
print(e.stats())
