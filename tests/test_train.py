"""CPU: the trainable V80 module carries the reference's parameter names and reproduces the reference model's outputs
(tests/golden/netfwd_splendor2_v80.npz was produced by the reference's own module); the trainer lowers its loss."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _load():
    from azg_amd.train import SplendorV80Module
    z = np.load(os.path.join(GOLDEN, 'weights_splendor2_v80.npz'))
    sd = {k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd/')}
    m = SplendorV80Module(num_players=2)
    m.load_state_dict(sd, strict=True)          # same keys as the reference checkpoint, nothing missing / unexpected
    return m.eval(), sd


def test_module_matches_reference_outputs_and_names():
    m, sd = _load()
    d = np.load(os.path.join(GOLDEN, 'netfwd_splendor2_v80.npz'))
    with torch.no_grad():
        lp, v = m(torch.from_numpy(d['boards']), torch.from_numpy(d['masks'].astype(bool)))
    assert np.allclose(torch.exp(lp).numpy(), d['pi'], atol=1e-5, rtol=0)
    assert np.allclose(v.numpy(), d['v'], atol=1e-5, rtol=0)
    assert set(m.state_dict().keys()) == set(sd.keys())


def test_losses_and_training_step():
    from azg_amd import train as T
    m, _ = _load()
    d = np.load(os.path.join(GOLDEN, 'netfwd_splendor2_v80.npz'))
    n = len(d['boards'])
    rng = np.random.default_rng(0)
    z = np.where(rng.random((n, 1)) < 0.5, 1.0, -1.0).astype(np.float32) * np.array([[1.0, -1.0]], dtype=np.float32)
    q = (0.5 * z).astype(np.float32)
    ex = (d['boards'].reshape(n, -1), d['pi'], z, d['masks'], q)
    # loss formulas, GenericNNetWrapper.py:179-191
    lp = torch.log(torch.tensor([[0.5, 0.5], [0.9, 0.1]]))
    t = torch.tensor([[1.0, 0.0], [0.5, 0.5]])
    kl = float(T.loss_pi(t, lp))
    ref = float((t * (torch.log(t.clamp_min(1e-30)) - lp)).sum() / 2)
    assert abs(kl - ref) < 1e-6
    lv = float(T.loss_v(torch.tensor([[1.0, -1.0]]), torch.tensor([[0.0, 0.0]]), torch.tensor([[0.0, 0.0]]), 1.0))
    assert abs(lv - (0.25 + 0.25) / 2) < 1e-7
    hist = T.train(m, ex, learn_rate=3e-3, batch_size=64, epochs=6, q_weight=0.5, device='cpu', seed=1)
    first, last = np.mean([h[1] for h in hist[:4]]), np.mean([h[1] for h in hist[-4:]])
    assert last < first                                   # the value head fits the synthetic targets
    assert not m.training


def test_module_4p_matches_reference_outputs():
    from azg_amd.train import SplendorV80Module
    z = np.load(os.path.join(GOLDEN, 'weights_splendor4_v80.npz'))
    m = SplendorV80Module(num_players=4)
    m.load_state_dict({k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd/')}, strict=True)
    d = np.load(os.path.join(GOLDEN, 'netfwd_splendor4_v80.npz'))
    with torch.no_grad():
        lp, v = m.eval()(torch.from_numpy(d['boards']), torch.from_numpy(d['masks'].astype(bool)))
    assert np.allclose(torch.exp(lp).numpy(), d['pi'], atol=1e-5, rtol=0) and np.allclose(v.numpy(), d['v'], atol=1e-5, rtol=0)


def test_azul_module_matches_reference_outputs_and_names():
    """AzulV84Module: the reference checkpoint loads with strict=True (same parameter names) and reproduces the reference
    model's golden outputs (BASELINE config 5 trains this net)."""
    from azg_amd.train import AzulV84Module
    z = np.load(os.path.join(GOLDEN, 'weights_azul_v84.npz'))
    sd = {k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith('sd/')}
    m = AzulV84Module()
    m.load_state_dict(sd, strict=True)
    d = np.load(os.path.join(GOLDEN, 'netfwd_azul_v84.npz'))
    with torch.no_grad():
        lp, v = m.eval()(torch.from_numpy(d['boards']), torch.from_numpy(d['masks'].astype(bool)))
    from test_nnet import assert_net_close
    assert_net_close(torch.exp(lp), v, 'azul_v84', d)


def _train_vs_reference(device):
    """train.train == the reference's GenericNNetWrapper.train (:44-92): same (pi loss, v loss) at both AdamW + OneCycleLR steps
    and the same weights afterwards (fixture: tools/gen_train_golden.py ran the reference's own trainer from
    pretrained_2players.pt on these 64 examples; one batch = all examples, so the sampling order does not matter)."""
    from azg_amd.train import train
    m, _ = _load()
    d = np.load(os.path.join(GOLDEN, 'train_splendor2_v80.npz'))
    ex = (d['boards'].reshape(len(d['boards']), -1), d['pi'], d['z'], d['valids'], d['q'])
    hist = train(m, ex, learn_rate=float(d['hp/learn_rate']), batch_size=int(d['hp/batch_size']), epochs=int(d['hp/epochs']),
                 q_weight=float(d['hp/q_weight']), device=device, seed=0)
    assert len(hist) == 2
    got_pi, got_v = np.array([h[0] for h in hist]), np.array([h[1] for h in hist])
    assert np.abs(got_pi - d['loss_pi']).max() <= 1e-5, (got_pi, d['loss_pi'])
    assert np.abs(got_v - d['loss_v']).max() <= 1e-5, (got_v, d['loss_v'])
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k in [f[6:] for f in d.files if f.startswith('after/')]:
        assert np.abs(sd[k] - d['after/' + k]).max() <= 1e-5, k


def test_trainer_matches_reference_trainer_cpu():
    _train_vs_reference('cpu')


import pytest  # noqa: E402


@pytest.mark.gpu
def test_trainer_matches_reference_trainer_gpu():
    _train_vs_reference('cuda:0')


def test_nnet_wrapper_examples_list_and_losses():
    """NNetWrapper.train takes Coach's example list (compressed pickles or tuples, Coach.py:84) and its loss_pi / loss_v are the
    reference's (:179-190): the first-step losses of the trainer fixture, evaluated directly."""
    import pickle
    import zlib
    from azg_amd.nnet_wrapper import decode_examples
    from azg_amd.train import loss_pi, loss_v
    d = np.load(os.path.join(GOLDEN, 'train_splendor2_v80.npz'))
    tuples = [(d['boards'][i], d['pi'][i], d['z'][i], d['valids'][i].astype(bool), d['q'][i]) for i in range(len(d['pi']))]
    packed = [zlib.compress(pickle.dumps(t), level=1) for t in tuples]
    for lst in (tuples, packed):
        cols = decode_examples(lst)
        assert np.array_equal(cols[0], d['boards'].reshape(len(tuples), -1)) and np.array_equal(cols[1], d['pi'])
    m, _ = _load()
    m.train()                                           # the reference evaluates the losses in training mode (BatchNorm batch stats)
    lp, v = m(torch.from_numpy(d['boards']), torch.from_numpy(d['valids'].astype(bool)))
    assert abs(float(loss_pi(torch.from_numpy(d['pi']), lp)) - d['loss_pi'][0]) <= 1e-5
    assert abs(float(loss_v(torch.from_numpy(d['z']), torch.from_numpy(d['q']), v, float(d['hp/q_weight']))) - d['loss_v'][0]) <= 1e-5
