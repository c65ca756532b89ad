"""worker of tests/test_gpu_coach.py::test_coach_learn_world2_equals_world1 -- one rank of a (possibly distributed) Coach.learn run.
usage: [torchrun ...] coach_worker.py OUTDIR   (WORLD_SIZE / RANK from the launcher; every rank uses cuda:0, collectives over gloo)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Args(dict):
    __getattr__ = dict.get


def main():
    out = sys.argv[1]
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from azg_amd import games
    from azg_amd.coach import Coach
    from azg_amd.train import SplendorV80Module
    g = games.SplendorGame(2)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'weights_splendor2_v80.npz'))
    m = SplendorV80Module(2)
    m.load_state_dict({k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd/')})
    ck = os.path.join(out, 'ckpt_w%d_r%d' % (world, rank))
    args = Args(numMCTSSims=8, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=False, dirichletAlpha=0.3, prob_fullMCTS=1.0,
                ratio_fullMCTS=5, temperature=[1.25, 0.8, 1.0], tempThreshold=6, numIters=2, numEps=13, numItersHistory=2,
                maxlenOfQueue=100000, learn_rate=1e-3, batch_size=64, epochs=1, q_weight=0.5, arenaCompare=6,
                updateThreshold=0.5, checkpoint=ck, seed=11)
    c = Coach(g, m, args, n_games=8, node_capacity=1024, log=lambda s: None)
    assert c.world == world and c.T_local == 8 // world
    res = c.learn()
    sd = {k: v.detach().cpu() for k, v in c.nnet.nnet.state_dict().items()}
    torch.save(dict(results=res, state_dict=sd), os.path.join(out, 'result_w%d_r%d.pt' % (world, rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
