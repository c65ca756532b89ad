"""Round time through a whole game: µs per round and descent levels per simulation in chunks of CHUNK rounds."""
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
e = SelfPlayEngine(g, net, a, T, node_capacity=16 * 800 + 512, max_examples=T * 160,
                   work_budget=int(os.environ.get('WB', '20')))
e.start()
total = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
prev = e.stats()
for c in range(total // chunk):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run(chunk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = e.stats()
    d = {k: s[k] - prev[k] for k in s if isinstance(s[k], (int, float)) and k in prev}
    sims = max(d.get('sims', 0), 1)
    print('rounds %6d  %6.1f us/round  %6.0f env-steps/s  levels/sim %.2f  plies %6d  games %5d  gc %5d  sims/round %.0f'
          % ((c + 1) * chunk, dt / chunk * 1e6, d.get('sims', 0) / 800 / dt, d.get('levels', 0) / sims, d.get('plies', 0),
             d.get('games', 0), d.get('gc_runs', 0), sims / chunk), flush=True)
    prev = s
print(e.stats())
