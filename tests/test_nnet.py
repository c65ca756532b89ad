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


@pytest.mark.gpu
@pytest.mark.parametrize('path', ['fused_net', 'fused_blocks', 'unfused'])
def test_v80_hip_kernels_forward_gpu(path):
    """The engine's own gfx950 net kernels vs the reference model's outputs: the 3-launch whole-net fusion
    (azg_nn_v80_forward), the per-block fusion (azg_nn_v80_block + skinny GEMMs) and the unfused kernel chain."""
    from azg_amd.nnet import SplendorV80Hip, SplendorV80
    root = os.path.join(os.path.dirname(__file__), 'golden')
    net = SplendorV80Hip.from_npz(os.path.join(root, 'weights_splendor2_v80.npz'), device='cuda:0', max_batch=512)
    net.fused_net = path == 'fused_net'
    net.fused_blocks = path != 'unfused'
    d = np.load(os.path.join(root, 'netfwd_splendor2_v80.npz'))
    boards = torch.from_numpy(d['boards']).cuda()
    masks = torch.from_numpy(d['masks']).cuda()
    pi, v = net.predict_batch(boards, masks)
    assert np.allclose(pi.cpu().numpy(), d['pi'], atol=1e-5, rtol=0)
    assert np.allclose(v.cpu().numpy(), d['v'], atol=1e-5, rtol=0)
    assert np.all(pi.cpu().numpy()[d['masks'] == 0] == 0)
    # ragged batch sizes (not a multiple of the 64-row GEMM tile / 8-sample SE group)
    ref = SplendorV80.from_npz(os.path.join(root, 'weights_splendor2_v80.npz'), device='cuda:0')
    for n in (1, 7, 65, 200):
        p1, v1 = net.predict_batch(boards[:n].contiguous(), masks[:n].contiguous())
        p2, v2 = ref.predict_batch(boards[:n], masks[:n])
        assert torch.allclose(p1, p2, atol=1e-5, rtol=0) and torch.allclose(v1, v2, atol=1e-5, rtol=0)


def _check_generic(cls, tag, device):
    from azg_amd import nnet
    root = os.path.join(os.path.dirname(__file__), 'golden')
    net = getattr(nnet, cls).from_npz(os.path.join(root, 'weights_%s.npz' % tag), device=device)
    d = np.load(os.path.join(root, 'netfwd_%s.npz' % tag))
    pi, v = net.predict_batch(torch.from_numpy(d['boards']).to(device), torch.from_numpy(d['masks']).to(device))
    # tolerance: 1e-5 on pi; 3e-5 on v -- an f64 evaluation of the Azul net differs from the reference's own f32 output by
    # 1.07e-5, i.e. 1e-5 is the reference's f32 rounding floor for this value head
    assert np.allclose(pi.cpu().numpy(), d['pi'], atol=1e-5, rtol=0)
    assert np.allclose(v.cpu().numpy(), d['v'], atol=3e-5, rtol=0)


@pytest.mark.parametrize('cls,tag', [('AzulV84', 'azul_v84'), ('SantoriniV89', 'santorini1_v89')])
def test_other_nets_forward_cpu(cls, tag):
    """azul/AzulNNet.py V84 and santorini/SantoriniNNet.py V89 re-expressed in plain torch vs the reference models."""
    _check_generic(cls, tag, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('cls,tag', [('AzulV84', 'azul_v84'), ('SantoriniV89', 'santorini1_v89')])
def test_other_nets_forward_gpu(cls, tag):
    _check_generic(cls, tag, 'cuda:0')
