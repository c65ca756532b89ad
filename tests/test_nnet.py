"""Net forward parity (G4): the plain-torch V80 re-expression vs the reference model's own outputs (fp32, <= 1e-5)."""
import os

import numpy as np
import pytest
import torch


def _check(device):
    from azg_amd.nnet import SplendorV80
    root = os.path.join(os.path.dirname(__file__), 'golden')
    net = SplendorV80.from_npz(os.path.join(root, 'weights_splendor2_v80.npz'), device=device)
    d = np.load(os.path.join(root, 'netfwd_splendor2_v80.npz'))
    pi, v = net.predict_batch(torch.from_numpy(d['boards']).to(device), torch.from_numpy(d['masks']).to(device))
    assert np.allclose(pi.cpu().numpy(), d['pi'], atol=1e-5, rtol=0)
    assert np.allclose(v.cpu().numpy(), d['v'], atol=1e-5, rtol=0)
    assert np.all(pi.cpu().numpy()[d['masks'] == 0] == 0)


def test_v80_forward_cpu():
    _check('cpu')


@pytest.mark.gpu
def test_v80_forward_gpu():
    _check('cuda:0')
