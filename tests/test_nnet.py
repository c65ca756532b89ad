"""Net forward parity (G4): the plain-torch V80 re-expression vs the reference model's own outputs (fp32, <= 1e-5)."""
import os

import numpy as np
import pytest
import torch


def assert_net_close(pi, v, tag, d=None):
    """<= 1e-5 against the reference model's outputs (G4, `netfwd_<tag>.npz` = the reference module's own f32 forward).  The
    value heads end in tanh(Linear(ReLU(Linear))) over a few hundred f32 products and the reference's f32 output itself sits
    up to 8.9e-6 (Azul) from the rounding-free value of its own forward (`netfwd64_<tag>.npz`: the same module evaluated with
    model.double(), tools/convert_ckpt.py --f64), so v is held to 1e-5 of that f64 value AND to the f32 output within 1e-5 plus
    the reference's own rounding distance, element by element."""
    root = os.path.join(os.path.dirname(__file__), 'golden')
    d = d if d is not None else np.load(os.path.join(root, 'netfwd_%s.npz' % tag))
    d64 = np.load(os.path.join(root, 'netfwd64_%s.npz' % tag))
    pi, v = pi.detach().cpu().numpy().astype(np.float64), v.detach().cpu().numpy().astype(np.float64)
    assert np.abs(pi - d['pi']).max() <= 1e-5, np.abs(pi - d['pi']).max()
    assert np.abs(pi - d64['pi64']).max() <= 1e-5
    assert np.abs(v - d64['v64']).max() <= 1e-5, np.abs(v - d64['v64']).max()
    assert np.all(np.abs(v - d['v']) <= 1e-5 + np.abs(d['v'] - d64['v64']))
    some = d['masks'].any(axis=1)           # (a board without a valid move -- ended games among the TLP boards: every logit is -1e8, pi uniform, as the reference's)
    assert np.all(pi[some][d['masks'][some] == 0] == 0)


def assert_close_on_random_boards(base, rb, rm, pi, v):
    """Random int8 boards are far outside the games' value ranges (activations 10-100x larger), so the f32 rounding floor of ANY
    f32 evaluation is above 1e-5 there: hold the kernel to 1e-5 of the f64 torch evaluation of the same weights plus the distance
    the plain f32 torch evaluation itself has from it, element by element."""
    p32, v32 = base.predict_batch(rb, rm)
    p32, v32 = p32.double(), v32.double()
    p64, v64 = base.to('cuda:0', torch.float64).predict_batch(rb, rm)
    p64, v64 = p64.double(), v64.double()
    assert bool(((pi.double() - p64).abs() <= 1e-5 + (p32 - p64).abs()).all())
    assert bool(((v.double() - v64).abs() <= 1e-5 + (v32 - v64).abs()).all())


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
@pytest.mark.parametrize('path', ['fused_net_h2', 'fused_net', 'fused_net_f32', 'fused_blocks', 'unfused'])
def test_v80_hip_kernels_forward_gpu(path):
    """The engine's own gfx950 net kernels vs the reference model's outputs: the 3-launch whole-net fusion
    (azg_nn_v80_forward), the per-block fusion (azg_nn_v80_block + skinny GEMMs) and the unfused kernel chain."""
    from azg_amd.nnet import SplendorV80Hip, SplendorV80
    root = os.path.join(os.path.dirname(__file__), 'golden')
    net = SplendorV80Hip.from_npz(os.path.join(root, 'weights_splendor2_v80.npz'), device='cuda:0', max_batch=512, split=path != 'fused_net_f32',
                                  h2=path == 'fused_net_h2')
    net.fused_net = path.startswith('fused_net')      # one launch: fp16 hi+lo operands on token-major tiles (default), bf16 x 3 expand, or f32 MFMAs throughout
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
    assert_net_close(pi, v, tag, d)


OTHER_NETS = [('AzulV84', 'azul_v84'), ('SantoriniV89', 'santorini1_v89'), ('SantoriniV78', 'santorini11_v78'),
              # the shipped nets of two f4 games are of the MobileNet-1d family: minivilles/pretrained_2players.pt (MinivillesNNet.py nn_version 82)
              # and thelittleprince/pretrained_3players.pt (TLPNNet.py nn_version 83); vectors from the reference's own modules
              ('MinivillesV82', 'minivilles2_v82'), ('TLPV83', 'tlp3_v83')]


@pytest.mark.parametrize('cls,tag', OTHER_NETS)
def test_other_nets_forward_cpu(cls, tag):
    """azul/AzulNNet.py V84 and santorini/SantoriniNNet.py V89 / V78 re-expressed in plain torch vs the reference models'
    own forward outputs (V78: SantoriniNNet.py:264-271 run on the unpickled pretrained_withgods.pt module; its torchvision
    InvertedResidual blocks execute tools/refshim's implementation of the published block algorithm)."""
    _check_generic(cls, tag, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('cls,tag', OTHER_NETS)
def test_other_nets_forward_gpu(cls, tag):
    _check_generic(cls, tag, 'cuda:0')


def _v78_module_restatement(sd):
    """Plain nn.Module restatement of SantoriniNNet nn_version 78 with the reference's parameter names (unfolded
    BatchNorm, nn layers): the independent fp32 implementation SantoriniV78 (folded, functional) is checked against.
    The reference's own module cannot run here (torchvision's InvertedResidual is absent) => parity unpinned for V78."""
    import torch.nn as nn

    class IR(nn.Module):
        def __init__(self):
            super().__init__()
            self.block = nn.Sequential(
                nn.Sequential(nn.Conv2d(64, 192, 1, bias=False), nn.BatchNorm2d(192), nn.ReLU()),
                nn.Sequential(nn.Conv2d(192, 192, 3, padding=1, groups=192, bias=False), nn.BatchNorm2d(192), nn.ReLU()),
                nn.Sequential(nn.Conv2d(192, 64, 1, bias=False), nn.BatchNorm2d(64)))

        def forward(self, x):
            return self.block(x) + x

    class Head(nn.Module):
        def __init__(self, bott, out, value):
            super().__init__()
            self.conv1x1, self.bn, self.value = nn.Conv2d(64, bott, 1, bias=False), nn.BatchNorm2d(bott), value
            if value:
                self.fc1, self.fc2 = nn.Linear(bott * 25 + 32, 64), nn.Linear(64, out)
            else:
                self.fc = nn.Linear(bott * 25 + 32, out)

        def forward(self, x, meta):
            x = torch.cat([torch.relu(self.bn(self.conv1x1(x))).flatten(1), meta], dim=1)
            return self.fc2(torch.relu(self.fc1(x))) if self.value else self.fc(x)

    class Net(nn.Module):
        def __init__(self, A):
            super().__init__()
            self.first_layer = nn.Conv2d(2, 64, 3, padding=1, bias=False)
            self.trunk = nn.Sequential(*[IR() for _ in range(10)])
            self.meta_fc = nn.Sequential(nn.Flatten(1), nn.Linear(25, 32), nn.ReLU())
            self.head_PI, self.head_V = Head(4, A, False), Head(2, 2, True)

        def forward(self, boards, valid):
            x = boards.permute(0, 3, 1, 2)
            f = self.trunk(self.first_layer(x[:, :2]))
            meta = self.meta_fc(x[:, 2:3])
            pi = torch.where(valid, self.head_PI(f, meta), torch.tensor(-1e8))
            return torch.exp(torch.log_softmax(pi, dim=1)), torch.tanh(self.head_V(f, meta))

    net = Net(sd['head_PI.fc.weight'].shape[0])
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items() if k != 'lowvalue'}, strict=True)
    return net.eval()


def _check_v78(device):
    from azg_amd.nnet import SantoriniV78
    root = os.path.join(os.path.dirname(__file__), 'golden')
    z = np.load(os.path.join(root, 'weights_santorini11_v78.npz'))
    sd = {k[3:]: z[k] for k in z.files if k.startswith('sd/')}
    env = np.load(os.path.join(root, 'env_santorini11.npz'))
    boards = torch.from_numpy(env['canonical'][:96].reshape(-1, 5, 5, 3).astype(np.int8))
    masks = torch.from_numpy(np.unpackbits(env['valid'][:96], axis=1, count=1782).astype(bool)) if env['valid'].shape[1] != 1782 \
        else torch.from_numpy(env['valid'][:96].astype(bool))
    ref = _v78_module_restatement(sd).double()               # f64: the rounding-free value of the restatement
    with torch.no_grad():
        pr, vr = ref(boards.double(), masks)
    net = SantoriniV78(sd, device=device)
    pi, v = net.predict_batch(boards.to(device), masks.to(device))
    assert pi.shape == (96, 1782) and v.shape == (96, 2)
    assert np.allclose(pi.cpu().numpy(), pr.numpy(), atol=1e-5, rtol=0)
    assert np.allclose(v.cpu().numpy(), vr.numpy(), atol=1e-5, rtol=0)
    assert np.all(pi.cpu().numpy()[~masks.numpy()] == 0)


def test_v78_forward_cpu():
    _check_v78('cpu')


@pytest.mark.gpu
def test_v78_forward_gpu():
    _check_v78('cuda:0')


def _check_v80_4p(device):
    """Splendor-4p checkpoint (pretrained_4players.pt, 88 tokens-channels): SplendorV80(num_players=4) vs the reference model"""
    from azg_amd.nnet import SplendorV80
    root = os.path.join(os.path.dirname(__file__), 'golden')
    net = SplendorV80.from_npz(os.path.join(root, 'weights_splendor4_v80.npz'), num_players=4, device=device)
    d = np.load(os.path.join(root, 'netfwd_splendor4_v80.npz'))
    pi, v = net.predict_batch(torch.from_numpy(d['boards']).to(device), torch.from_numpy(d['masks']).to(device))
    assert v.shape[1] == 4
    assert np.allclose(pi.cpu().numpy(), d['pi'], atol=1e-5, rtol=0)
    assert np.allclose(v.cpu().numpy(), d['v'], atol=1e-5, rtol=0)


def test_v80_4p_forward_cpu():
    _check_v80_4p('cpu')


@pytest.mark.gpu
def test_v80_4p_forward_gpu():
    _check_v80_4p('cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [False, True, 'h2'], ids=['launches17', 'fused_f32', 'fused_h2'])
@pytest.mark.parametrize('tag', ['splendor2_v80', 'splendor4_v80', 'azul_v84', 'minivilles2_v82', 'tlp3_v83'])
def test_mobilenet1d_engine_kernels_gpu(tag, fused):
    """MobileNet1dHip (engine GEMM / depthwise / head kernels, any geometry) vs the reference models' golden outputs, and at
    a batch that is not a multiple of the 16-row tiles vs the torch-ops evaluation of the same weights."""
    from azg_amd import nnet
    root = os.path.join(os.path.dirname(__file__), 'golden')
    if tag == 'azul_v84':
        base = nnet.AzulV84.from_npz(os.path.join(root, 'weights_%s.npz' % tag), device='cuda:0')
    elif tag in ('minivilles2_v82', 'tlp3_v83'):
        base = nnet.MobileNet1d.from_npz(os.path.join(root, 'weights_%s.npz' % tag), device='cuda:0')
    else:
        npl = 4 if tag.startswith('splendor4') else 2
        base = nnet.SplendorV80.from_npz(os.path.join(root, 'weights_%s.npz' % tag), num_players=npl, device='cuda:0')
    if not fused and tag in ('minivilles2_v82', 'tlp3_v83'):
        with pytest.raises(ValueError):            # the launch-per-layer path exists for the Splendor / Azul geometries only
            nnet.MobileNet1dHip(base, max_batch=64, fused=False)
        return
    # one launch with the GEMM phases on f16 x 2 split-precision operands (default) / on f32 MFMAs / 17 launches
    net = nnet.MobileNet1dHip(base, max_batch=64, fused=bool(fused), h2=fused == 'h2')
    assert net.fused == bool(fused)
    d = np.load(os.path.join(root, 'netfwd_%s.npz' % tag))
    boards = torch.from_numpy(d['boards']).to('cuda:0')
    masks = torch.from_numpy(d['masks']).to('cuda:0')
    pi, v = net.predict_batch(boards.to(torch.int8), masks)
    assert_net_close(pi, v, tag, d)
    # ragged batch (B * L not a multiple of 16), larger than max_batch (buffers regrow), random boards
    g = torch.Generator().manual_seed(5)
    B = 203
    rb = torch.randint(0, 6, (B,) + tuple(boards.shape[1:]), generator=g, dtype=torch.int8).to('cuda:0')
    rm = (torch.rand((B, masks.shape[1]), generator=g) < 0.4).to('cuda:0')
    rm[:, -1] = True
    pi2, v2 = net.predict_batch(rb, rm)
    assert_close_on_random_boards(base, rb, rm, pi2, v2)
    assert float(pi2[~rm].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('split', ['h2', True, False])
def test_santorini_v89_one_launch_gpu(split):
    """SantoriniV89Hip (implicit-GEMM 3x3 convolutions + heads in one launch) vs the reference model's golden outputs and,
    on a ragged random batch, vs the MIOpen evaluation of the same weights."""
    from azg_amd import nnet
    root = os.path.join(os.path.dirname(__file__), 'golden')
    base = nnet.SantoriniV89.from_npz(os.path.join(root, 'weights_santorini1_v89.npz'), device='cuda:0')
    # trunk on f16 x 2 split-precision operands (default) / bf16 x 3 / f32 MFMA
    net = nnet.SantoriniV89Hip(base, max_batch=64, split=split is True, h2=split == 'h2')
    d = np.load(os.path.join(root, 'netfwd_santorini1_v89.npz'))
    boards = torch.from_numpy(d['boards']).to('cuda:0').to(torch.int8)
    masks = torch.from_numpy(d['masks']).to('cuda:0')
    pi, v = net.predict_batch(boards, masks)
    assert_net_close(pi, v, 'santorini1_v89', d)
    g = torch.Generator().manual_seed(9)
    B = 203
    rb = torch.randint(-2, 5, (B, 5, 5, 3), generator=g, dtype=torch.int8).to('cuda:0')
    rm = (torch.rand((B, 162), generator=g) < 0.3).to('cuda:0')
    rm[:, 0] = True
    pi2, v2 = net.predict_batch(rb, rm)
    assert_close_on_random_boards(base, rb, rm, pi2, v2)
    assert float(pi2[~rm].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('split', ['h2', False, True])
def test_santorini_v78_one_launch_gpu(split):
    """SantoriniV78Hip (trunk launch: MFMA 1x1 convolutions -- f32 or split-precision bf16 x 3 --, in-place depthwise 3x3, value head;
    policy FC launch) vs the reference model's own
    forward outputs (netfwd_santorini11_v78.npz: SantoriniNNet.forward on pretrained_withgods.pt), then vs the torch-ops
    SantoriniV78 of the same weights on more boards from the golden env trajectories and on a ragged random batch."""
    from azg_amd import nnet
    root = os.path.join(os.path.dirname(__file__), 'golden')
    base = nnet.SantoriniV78.from_npz(os.path.join(root, 'weights_santorini11_v78.npz'), device='cuda:0')
    net = nnet.SantoriniV78Hip(base, max_batch=64, split=split is True, h2=split == 'h2')    # f16 x 2 (default) / f32 / bf16 x 3 trunk
    d = np.load(os.path.join(root, 'netfwd_santorini11_v78.npz'))
    pi, v = net.predict_batch(torch.from_numpy(d['boards']).to('cuda:0').to(torch.int8).reshape(-1, 75),
                              torch.from_numpy(d['masks']).to('cuda:0'))
    assert_net_close(pi, v, 'santorini11_v78', d)
    env = np.load(os.path.join(root, 'env_santorini11.npz'))
    states = torch.from_numpy(env['state'][:150].astype(np.int8)).reshape(-1, 75).to('cuda:0')
    g = torch.Generator().manual_seed(3)
    base64 = nnet.SantoriniV78.from_npz(os.path.join(root, 'weights_santorini11_v78.npz'), device='cuda:0', dtype=torch.float64)
    for boards in (states, torch.randint(-2, 5, (203, 75), generator=g, dtype=torch.int8).to('cuda:0')):
        B = boards.shape[0]
        rm = (torch.rand((B, 1782), generator=g) < 0.05).to('cuda:0')
        rm[:, 7] = True
        pi, v = net.predict_batch(boards, rm)
        pr, vr = base64.predict_batch(boards.reshape(B, 5, 5, 3), rm)
        assert float((pi - pr).abs().max()) < 1e-5 and float((v - vr).abs().max()) < 1e-5
        assert float(pi[~rm].abs().max()) == 0.0 and abs(float(pi.sum(dim=1).mean()) - 1.0) < 1e-5


@pytest.mark.gpu
def test_santorini_v78_policy_in_one_launch_gpu():
    """AZG_S78_POLICY2=0: the with-gods net's policy FC + softmax as ONE launch (k_s78_policy_h2, 16 samples per workgroup, no workspace) instead
    of the default GEMM + softmax pair -- the switch is read once per process, hence the subprocess: same outputs within 1e-5 of the f64
    evaluation, exact zeros on invalid actions, on a ragged batch."""
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from azg_amd import nnet
root = %r
w = os.path.join(root, 'weights_santorini11_v78.npz')
net = nnet.SantoriniV78Hip(nnet.SantoriniV78.from_npz(w, device='cuda:0'), max_batch=64)
ref = nnet.SantoriniV78.from_npz(w, device='cuda:0', dtype=torch.float64)
g = torch.Generator().manual_seed(5)
boards = torch.randint(-2, 5, (203, 75), generator=g, dtype=torch.int8).to('cuda:0')
rm = (torch.rand((203, 1782), generator=g) < 0.05).to('cuda:0'); rm[:, 7] = True
pi, v = net.predict_batch(boards, rm)
pr, vr = ref.predict_batch(boards.reshape(203, 5, 5, 3), rm)
assert float((pi - pr).abs().max()) < 1e-5 and float((v - vr).abs().max()) < 1e-5
assert float(pi[~rm].abs().max()) == 0.0
print('ok')
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    env = dict(os.environ, AZG_S78_POLICY2='0')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['splendor2_v80', 'splendor4_v80', 'azul_v84', 'santorini1_v89', 'santorini11_v78', 'minivilles2_v82', 'tlp3_v83'])
def test_net_kernels_do_not_depend_on_stale_onchip_memory(tag):
    """every one-launch net kernel gives bit-identical outputs whatever the LDS of the CUs and the scratch memory of the queue held
    before (azg_debug_poison_onchip): no read of on-chip memory the kernel did not write"""
    from azg_amd import nnet
    from conftest import poison_onchip
    root = os.path.join(os.path.dirname(__file__), 'golden')
    w = os.path.join(root, 'weights_%s.npz' % tag)
    if tag == 'splendor2_v80':
        net = nnet.SplendorV80Hip.from_npz(w, device='cuda:0', max_batch=256)
    elif tag == 'splendor4_v80':
        net = nnet.MobileNet1dHip(nnet.SplendorV80.from_npz(w, num_players=4, device='cuda:0'), max_batch=256)
    elif tag == 'azul_v84':
        net = nnet.MobileNet1dHip(nnet.AzulV84.from_npz(w, device='cuda:0'), max_batch=256)
    elif tag in ('minivilles2_v82', 'tlp3_v83'):
        net = nnet.MobileNet1dHip(nnet.MobileNet1d.from_npz(w, device='cuda:0'), max_batch=256)
    elif tag == 'santorini1_v89':
        net = nnet.SantoriniV89Hip(nnet.SantoriniV89.from_npz(w, device='cuda:0'), max_batch=256)
    else:
        net = nnet.SantoriniV78Hip(nnet.SantoriniV78.from_npz(w, device='cuda:0'), max_batch=256)
    d = np.load(os.path.join(root, 'netfwd_%s.npz' % tag))
    reps = -(-203 // len(d['boards']))
    boards = torch.from_numpy(np.concatenate([d['boards']] * reps)[:203]).to('cuda:0').to(torch.int8)
    masks = torch.from_numpy(np.concatenate([d['masks']] * reps)[:203]).to('cuda:0')
    outs = []
    for pattern in (0x0, 0xFFFFFFFF, 0x7FC00000, 0xA5A5A5A5):
        poison_onchip(pattern)
        pi, v = net.predict_batch(boards, masks)
        outs.append((pi.clone(), v.clone()))
    for pi, v in outs[1:]:
        assert torch.equal(pi, outs[0][0]) and torch.equal(v, outs[0][1])


@pytest.mark.gpu
def test_h2_kernels_saturate_out_of_range_activations_gpu():
    """Activation range of the f16 x 2 kernels (include/azg.h, 'Activation range'): their LDS planes hold 64 * x as f16, so |x| >= 1023.5
    does not fit.  The kernels run with the FP16_OVFL mode bit set: such a value saturates -- pi / v stay finite, never NaN -- and the
    f32-operand kernel of the same net (h2=False, split=False) has the full range and stays on the fp32 model.  A net whose first layer
    is scaled up 2000 x drives the residual stream to ~1e4 (GenericNNetWrapper.train can diverge like this)."""
    from azg_amd import nnet
    root = os.path.join(os.path.dirname(__file__), 'golden')
    z = np.load(os.path.join(root, 'weights_splendor2_v80.npz'))
    sd = {k[3:]: torch.as_tensor(z[k]).clone() for k in z.files if k.startswith('sd/')}
    sd['first_layer.linear.weight'] *= 2000.0
    torch.manual_seed(5)
    B = 64
    boards = torch.randint(0, 6, (B, 56, 7), dtype=torch.int8, device='cuda:0')
    valids = (torch.rand((B, 81), device='cuda:0') < 0.5).to(torch.uint8)
    valids[:, 80] = 1
    ref = nnet.SplendorV80(sd, device='cuda:0', dtype=torch.float64)
    x0 = torch.matmul(boards.reshape(B, 56, 7).double().transpose(1, 2), ref.W0) + ref.b0
    assert float(x0.abs().max()) > 2000.0                      # the tile really leaves the planes' range
    p64, v64 = ref.predict_batch(boards, valids.bool())
    h2 = nnet.SplendorV80Hip(sd, device='cuda:0', max_batch=B, h2=True)
    pi, v = h2.predict_batch(boards, valids)
    assert bool(torch.isfinite(pi).all()) and bool(torch.isfinite(v).all()), 'out-of-range activations must saturate, not turn into inf / NaN'
    assert torch.allclose(pi.sum(dim=1), torch.ones(B, device='cuda:0'), atol=1e-4)
    f32 = nnet.SplendorV80Hip(sd, device='cuda:0', max_batch=B, h2=False, split=False)
    pf, vf = f32.predict_batch(boards, valids)
    assert float((pf.double() - p64).abs().max()) < 1e-3 and float((vf.double() - v64).abs().max()) < 1e-3     # f32 arithmetic at 1e4 magnitudes
    # the generic kernel splits its f32 tiles as they are read; a saturated operand may not meet zero padding as inf * 0 = NaN
    sd4 = {k[3:]: torch.as_tensor(np.load(os.path.join(root, 'weights_azul_v84.npz'))[k]).clone()
           for k in np.load(os.path.join(root, 'weights_azul_v84.npz')).files if k.startswith('sd/')}
    key = [k for k in sd4 if k.startswith('first_layer') and k.endswith('linear.weight')][0]
    sd4[key] *= 5000.0
    base = nnet.AzulV84(sd4, device='cuda:0')
    mb = nnet.MobileNet1dHip(base, max_batch=B, h2=True)
    ab = torch.randint(0, 5, (B, 23, 6), dtype=torch.int8, device='cuda:0')
    av = (torch.rand((B, base.A), device='cuda:0') < 0.5).to(torch.uint8)
    av[:, 0] = 1
    pa, va = mb.predict_batch(ab, av)
    assert bool(torch.isfinite(pa).all()) and bool(torch.isfinite(va).all())
