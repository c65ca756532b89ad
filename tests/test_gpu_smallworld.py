"""GPU parity of the Smallworld plugin's random symmetries: get_symmetries shifts both scores by two
np.random.randint offsets (SmallworldLogicNumba.py:281-299); the engine draws them from the counter streams the reference drew from
(tools/gen_golden_smallworld.py).  Everything else of the plugin runs through the standard parametrised tests (test_gpu_env / _mcts /
_selfplay with variant 'smallworld')."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [2, 3, 4])
def test_smallworld_random_symmetries_vs_golden(golden_dir, n):
    from azg_amd import games
    d = np.load(os.path.join(golden_dir, 'sym_smallworld%s.npz' % ('' if n == 2 else n)))
    g = games.SmallworldGame(n)
    dev = g.device
    assert g.max_symmetries() == 3 and (g.S, g.A, g.P) == ({2: 320, 3: 416, 4: 528}[n], {2: 131, 3: 166, 4: 211}[n], n)
    ob, op, ov, cnt = g.symmetries_batch(torch.from_numpy(d['state']).to(dev), torch.from_numpy(d['pi']).to(dev),
                                         torch.from_numpy(d['valids']).to(dev), rng_seed=int(d['seed']), stream0=0)
    assert np.array_equal(cnt.cpu().numpy(), d['count'])
    assert np.array_equal(ob.cpu().numpy(), d['out_state']) and np.array_equal(op.cpu().numpy(), d['out_pi'])
    assert np.array_equal(ov.cpu().numpy(), d['out_valids'])
    b0 = d['state'][3].reshape(g.getBoardSize())
    syms = g.getSymmetries(b0, d['pi'][3], d['valids'][3].astype(bool))
    assert len(syms) == 3 and np.array_equal(syms[0][0], b0) and not np.array_equal(syms[1][0], b0)
