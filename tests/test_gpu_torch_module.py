"""The evaluator of games without an engine net: any torch module with the reference's forward signature runs as
the leaf evaluator on PyTorch-ROCm (nnet.TorchModuleEvaluator) behind the same NNetWrapper / SelfPlayEngine / Coach surface -- here a
small module on The Little Prince: predict parity with the module itself, a captured self-play run to the episode quota, one
training pass, checkpoint round trip."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


class Args(dict):
    __getattr__ = dict.get


class TinyNet(nn.Module):
    """forward(board f32[B, 55, 15], valid bool[B, 9]) -> (log_softmax of the masked logits, tanh value[B, 3]): the reference's contract"""

    def __init__(self, shape, A, P):
        super().__init__()
        n = int(np.prod(shape))
        self.body = nn.Linear(n, 64)
        self.pi, self.v = nn.Linear(64, A), nn.Linear(64, P)

    def forward(self, board, valid_actions):
        assert board.dtype == torch.float32 and valid_actions.dtype == torch.bool and board.dim() == 3
        h = F.relu(self.body(board.flatten(1) / 16.0))
        logits = torch.where(valid_actions, self.pi(h), torch.full_like(self.pi(h), -1e8))
        return F.log_softmax(logits, dim=1), torch.tanh(self.v(h))


def test_torch_module_behind_the_plugin_surface(tmp_path):
    from azg_amd import games
    from azg_amd.nnet_wrapper import NNetWrapper
    from azg_amd.selfplay import SelfPlayEngine
    torch.manual_seed(3)
    g = games.TLPGame(3, rng_seed=11)
    mod = TinyNet(g.getBoardSize(), g.A, g.P)
    w = NNetWrapper(g, dict(nn_version=-1, learn_rate=1e-3, batch_size=64, epochs=1), module=mod)
    b = g.getInitBoard()
    va = g.getValidMoves(b, 0)
    pi, v = w.predict(b, va)
    with torch.no_grad():
        lp, vv = mod(torch.from_numpy(b[None].astype(np.float32)).to(g.device), torch.from_numpy(va[None]).to(g.device))
    assert np.allclose(pi, torch.exp(lp)[0].cpu().numpy(), atol=1e-6) and np.allclose(v, vv[0].cpu().numpy(), atol=1e-6)
    assert pi[~va].max() < 1e-12 and abs(pi.sum() - 1) < 1e-5
    T = 32
    args = Args(numMCTSSims=24, cpuct=1.0, fpu=0.0, universes=1, forced_playouts=False, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0,
                temperature=[1.25, 0.8, 1.0], tempThreshold=6)
    eng = SelfPlayEngine(g, w.evaluator(T), args, n_games=T, node_capacity=2048, max_examples=T * 100)     # HIP-graph rounds
    eng.start(episode_quota=T)
    for _ in range(400):
        eng.run(rounds=160)
        if eng.stats()['active'] == 0:
            break
    st = eng.stats()
    assert st['errors'] == 0 and st['games'] == T, st
    ex = eng.drain_examples(symmetries=True)
    assert ex[0].shape[0] >= T * 48 and torch.isfinite(ex[1]).all()
    before = {k: t.clone() for k, t in mod.state_dict().items()}
    hist = w.train([x.cpu().numpy() for x in ex[:5]], seed=1)
    assert any(not torch.equal(before[k].to(t.device), t) for k, t in mod.state_dict().items()) and hist is not None
    w.save_checkpoint(str(tmp_path), 'tiny.pt', additional_keys={'nn_version': -1})
    w2 = NNetWrapper(g, dict(nn_version=-1), module=TinyNet(g.getBoardSize(), g.A, g.P))
    assert w2.load_checkpoint(str(tmp_path), 'tiny.pt') is not None
    pi2, _ = w2.predict(b, va)
    pi1, _ = w.predict(b, va)
    assert np.allclose(pi1, pi2, atol=1e-6)


def test_coach_learn_with_a_custom_module(tmp_path):
    """Coach.learn with a caller-supplied module (no engine net for this game): the competitor network is a copy of the module
    (Coach.py:30 rebuilds it from args, which cannot know the module), training sees boards in the game's shape, the arena gate runs"""
    from azg_amd import games
    from azg_amd.coach import Coach
    from azg_amd.nnet_wrapper import NNetWrapper
    torch.manual_seed(5)
    g = games.TLPGame(3, rng_seed=13)
    for bare in (False, True):
        mod = TinyNet(g.getBoardSize(), g.A, g.P)
        nnet = mod if bare else NNetWrapper(g, dict(nn_version=-1, learn_rate=1e-3, batch_size=64, epochs=1), module=mod)
        args = Args(numMCTSSims=16, cpuct=1.0, fpu=0.0, universes=1, forced_playouts=False, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0,
                    temperature=[1.25, 0.8, 1.0], tempThreshold=6, numIters=1, numEps=8, numItersHistory=2, maxlenOfQueue=100000, learn_rate=1e-3,
                    batch_size=64, epochs=1, q_weight=0.5, arenaCompare=4, updateThreshold=0.5, checkpoint=str(tmp_path / ('bare' if bare else 'wrapped')))
        c = Coach(g, nnet, args, n_games=8, node_capacity=1024, log=lambda s: None)
        assert c.pnet.nnet is not c.nnet.nnet and isinstance(c.pnet.nnet, TinyNet) and c.nnet._custom and c.pnet._custom
        res = c.learn()
        assert len(res) == 1 and res[0]['nwins'] + res[0]['pwins'] + res[0]['draws'] == 4 and res[0]['examples'] > 32
