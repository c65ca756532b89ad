"""Does the asynchronous pipeline survive a process that has created many streams?  (HIP multiplexes streams onto a few hardware
queues; the pipeline's two kernels must not share one.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from azg_amd import games
from azg_amd.nnet import SplendorV80Hip
from azg_amd.selfplay import SelfPlayEngine


class Args(dict):
    __getattr__ = dict.get


n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 12
streams = [torch.cuda.Stream() for _ in range(n_streams)]
x = torch.zeros(1024, device='cuda')
for s in streams:
    with torch.cuda.stream(s):
        x = x + 1
torch.cuda.synchronize()
g = games.SplendorGame(2)
w = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'weights_splendor2_v80.npz')
args = Args(numMCTSSims=24, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6,
            cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True)
for use_side in (False, True):
    T = 40
    net = SplendorV80Hip.from_npz(w, max_batch=T)
    e = SelfPlayEngine(g, net, args, T, node_capacity=2048, max_examples=T * 400, rng_seed=9, async_pipe=True, async_cfg=dict(n_net=3, n_sel=5))
    e.start()
    if use_side:
        with torch.cuda.stream(streams[3]):
            e.run(200)
    else:
        e.run(200)
    torch.cuda.synchronize()
    st = e.stats()
    print('streams created %d, caller on %s: errors %d plies %d ctl %s' % (n_streams, 'a side stream' if use_side else 'the default stream',
                                                                          st['errors'], st['plies'], e.forest.async_profile()['ctl']))
    e.close()
