"""Long self-play soak at the benchmark configuration: whole games incl. restarts and GC; checks the engine error flags,
the structural validator and that finished games produced examples.   python tools/soak.py [rounds]"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
T = 4096
g = games.SplendorGame(2)
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=T)
WB = int(os.environ.get('WB', '20')); CAP = int(os.environ.get('CAP', '13312'))
e = SelfPlayEngine(g, net, a, T, node_capacity=CAP, max_examples=T * 160, work_budget=WB)
print('work_budget', WB, 'cap', CAP)
e.start()
t0 = time.time()
done = 0
prev = e.stats(); tp = time.time()
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
while done < rounds:
    e.run(CH); done += CH
    s = e.stats(); tn = time.time()
    ds = s['sims'] - prev['sims']
    print('rounds %6d  %.3f ms/round  yield %.2f  levels/sim %.2f  plies %d games %d examples %d gc_runs %d (+%d) max_live %d errors %d' % (
        done, (tn - tp) / CH * 1e3, ds / CH / T, (s['levels'] - prev['levels']) / max(1, ds), s['plies'], s['games'], s['examples'],
        s['gc_runs'], s['gc_runs'] - prev['gc_runs'], s['max_live_after_gc'], s['errors']), flush=True)
    prev, tp = s, tn
    if s['errors']:
        break
bad = e.forest.validate()
ex = e.drain_examples()
print('validate() ->', bad, ' drained examples', ex[0].shape[0], ' env-steps/s %.0f' % (s['plies'] / (time.time() - t0)))
assert bad == 0 and ex[0].shape[0] > 0 and s['errors'] == 0
