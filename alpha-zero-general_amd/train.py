"""Next-row f1: the trainer side of Coach.learn on PyTorch-ROCm autograd.

`SplendorV80Module` is a trainable nn.Module with the reference's parameter names (splendor/SplendorNNet.py:148-202,
262-283,397-440), so `state_dict()` round-trips with the reference's checkpoints and with the inference nets of
azg_amd.nnet (which fold its BatchNorms).  `train()` is GenericNNetWrapper.train (:44-92): AdamW + OneCycleLR,
loss = KLDiv(pi) + 0.25 * MSE((z + q_weight*q) / (1 + q_weight)) (:179-190), batches sampled without replacement inside a
batch, `epochs * (len(examples) // batch_size)` steps."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class _LinearNormAct(nn.Module):
    """Linear over the channel axis (or over the token axis when depthwise) of x[B, C, L] + BatchNorm1d(C) + activation"""

    def __init__(self, n_in, n_out, act, depthwise=False, channels=None):
        super().__init__()
        self.linear = nn.Linear(n_in, n_out, bias=False)
        self.norm = nn.BatchNorm1d(channels if depthwise else n_out)
        self.activation = act() if act is not None else nn.Identity()
        self.depthwise = depthwise

    def forward(self, x):
        y = self.linear(x) if self.depthwise else self.linear(x.transpose(1, 2)).transpose(1, 2)
        return self.activation(self.norm(y))


class _SE(nn.Module):
    def __init__(self, channels, squeeze, setype):
        super().__init__()
        self.setype = setype
        self.fc1, self.fc2 = nn.Linear(channels, squeeze), nn.Linear(squeeze, channels)

    def forward(self, x):
        s = x.mean(dim=2) if self.setype == 'avg' else x.amax(dim=2)
        s = F.hardsigmoid(self.fc2(F.relu(self.fc1(s))))
        return x * s[:, :, None]


def _make_divisible(v, divisor=8, min_value=None):
    """torchvision.models._utils._make_divisible (the SE squeeze width of the reference blocks)"""
    min_value = divisor if min_value is None else min_value
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    return new_v + divisor if new_v < 0.9 * v else new_v


class _Block(nn.Module):
    """InvertedResidual1d (SplendorNNet.py:189-202, AzulNNet.py:47-80): residual only when in == out channels"""

    def __init__(self, c, e, use_hs, setype, tokens=7, c_out=None):
        super().__init__()
        c_out = c if c_out is None else c_out
        act = nn.Hardswish if use_hs else nn.ReLU
        self.expand = _LinearNormAct(c, e, act)
        self.depthwise = _LinearNormAct(tokens, tokens, act, depthwise=True, channels=e)
        self.se = _SE(e, _make_divisible(e // 4, 8), setype)
        self.project = _LinearNormAct(e, c_out, None)
        self.use_res_connect = c == c_out

    def forward(self, x):
        y = self.project(self.se(self.depthwise(self.expand(x))))
        return y + x if self.use_res_connect else y


class SplendorV80Module(nn.Module):
    version = 80

    def __init__(self, num_players=2, action_size=81, dropout=0.0):
        super().__init__()
        self.C = 32 + 10 * num_players + num_players * num_players
        self.P, self.A, self.dropout = num_players, action_size, dropout
        C = self.C
        self.first_layer = _LinearNormAct(C, C, None)
        self.trunk = nn.Sequential(_Block(C, 3 * C, False, 'avg'))
        self.output_layers_PI = nn.Sequential(_Block(C, 3 * C, True, 'max'), nn.Flatten(1), nn.Linear(7 * C, action_size), nn.ReLU(),
                                              nn.Linear(action_size, action_size))
        self.output_layers_V = nn.Sequential(_Block(C, 3 * C, True, 'max'), nn.Flatten(1), nn.Linear(7 * C, num_players), nn.ReLU(),
                                             nn.Linear(num_players, num_players))
        self.register_buffer('lowvalue', torch.FloatTensor([-1e8]))

    def forward(self, boards, valid_actions):
        """-> (log pi [B, A], v [B, P]) like the reference module"""
        x = boards.reshape(-1, self.C, 7).float()
        x = self.first_layer(x)
        x = F.dropout(self.trunk(x), p=self.dropout, training=self.training)
        v = self.output_layers_V(x)
        pi = torch.where(valid_actions.bool(), self.output_layers_PI(x), self.lowvalue)
        return F.log_softmax(pi, dim=1), torch.tanh(v)


class AzulV84Module(nn.Module):
    """azul/AzulNNet.py nn_version 84 (:91-113,130-142) with the reference's parameter names: [B, 23, 6] boards, trunk block
    23->115->23, policy head block 23->115->46 (no residual) + Linear(276,180)+ReLU+Linear, value head block 23->46->23."""
    version = 84

    def __init__(self, num_players=2, action_size=180, dropout=0.0):
        super().__init__()
        self.C, self.L, self.P, self.A, self.dropout = 23, 6, num_players, action_size, dropout
        C, L = self.C, self.L
        self.first_layer = _LinearNormAct(C, C, None)
        self.trunk = nn.Sequential(_Block(C, 5 * C, False, 'avg', tokens=L))
        self.output_layers_PI = nn.Sequential(_Block(C, 5 * C, True, 'avg', tokens=L, c_out=2 * C), nn.Flatten(1),
                                              nn.Linear(2 * C * L, action_size), nn.ReLU(), nn.Linear(action_size, action_size))
        self.output_layers_V = nn.Sequential(_Block(C, 2 * C, True, 'avg', tokens=L), nn.Flatten(1), nn.Linear(C * L, num_players),
                                             nn.ReLU(), nn.Linear(num_players, num_players))
        self.register_buffer('lowvalue', torch.FloatTensor([-1e8]))

    def forward(self, boards, valid_actions):
        x = boards.reshape(-1, self.C, self.L).float()
        x = self.first_layer(x)
        x = F.dropout(self.trunk(x), p=self.dropout, training=self.training)
        v = self.output_layers_V(x)
        pi = torch.where(valid_actions.bool(), self.output_layers_PI(x), self.lowvalue)
        return F.log_softmax(pi, dim=1), torch.tanh(v)


class _ResBlock(nn.Module):
    """SimpleResBlock (santorini/SantoriniNNet.py:71-84): conv3x3+BN+ReLU, conv3x3+BN, + input, ReLU"""

    def __init__(self, c):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c)
        self.conv2, self.bn2 = nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(y)) + x)


class _Head2d(nn.Module):
    """SimpleHead / HeadWithMeta (SantoriniNNet.py:17-69): 1x1 bottleneck conv + BN + ReLU, flatten (+ the 32 metadata
    features), Linear (policy) or Linear + ReLU + Linear (value)"""

    def __init__(self, c, bottleneck, out, value, meta=0):
        super().__init__()
        self.conv1x1, self.bn, self.value = nn.Conv2d(c, bottleneck, 1, bias=False), nn.BatchNorm2d(bottleneck), value
        flat = bottleneck * 25 + meta
        if value:
            self.fc1, self.fc2 = nn.Linear(flat, 64), nn.Linear(64, out)
        else:
            self.fc = nn.Linear(flat, out)

    def forward(self, x, meta=None):
        x = torch.flatten(F.relu(self.bn(self.conv1x1(x))), 1)
        if meta is not None:
            x = torch.cat([x, meta], dim=1)
        return self.fc2(F.relu(self.fc1(x))) if self.value else self.fc(x)


class SantoriniV89Module(nn.Module):
    """santorini/SantoriniNNet.py nn_version 89 (:194-219,273-281; no gods, A = 162) with the reference's parameter names:
    conv3x3(2->64)+BN+ReLU, five SimpleResBlocks, SimpleHead heads (bottlenecks 2 / 1)."""
    version = 89

    def __init__(self, num_players=2, action_size=162, dropout=0.0):
        super().__init__()
        self.P, self.A, self.dropout = num_players, action_size, dropout
        self.first_layer = nn.Sequential(nn.Conv2d(2, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64), nn.ReLU())
        self.trunk = nn.Sequential(*[_ResBlock(64) for _ in range(5)])
        self.head_PI, self.head_V = _Head2d(64, 2, action_size, False), _Head2d(64, 1, num_players, True)
        self.register_buffer('lowvalue', torch.FloatTensor([-1e8]))

    def forward(self, boards, valid_actions):
        x = boards.reshape(-1, 5, 5, 3).float().permute(0, 3, 1, 2)
        f = self.trunk(self.first_layer(x[:, :2]))
        v = self.head_V(f)
        pi = torch.where(valid_actions.bool(), self.head_PI(f), self.lowvalue)
        return F.log_softmax(pi, dim=1), torch.tanh(v)


class _MBBlock2d(nn.Module):
    """torchvision's MobileNetV3 InvertedResidual as the reference configures it (SantoriniNNet.py:172-178: 64 -> 192 -> 64,
    3x3 depthwise, no SE, ReLU, residual); parameter names block.{0,1,2}.{0 conv, 1 BatchNorm}"""

    def __init__(self, c, e):
        super().__init__()
        cna = lambda i, o, k, g, act: nn.Sequential(nn.Conv2d(i, o, k, padding=(k - 1) // 2, groups=g, bias=False),  # noqa: E731
                                                    nn.BatchNorm2d(o), *([nn.ReLU()] if act else []))
        self.block = nn.Sequential(cna(c, e, 1, 1, True), cna(e, e, 3, e, True), cna(e, c, 1, 1, False))

    def forward(self, x):
        return self.block(x) + x


class SantoriniV78Module(nn.Module):
    """santorini/SantoriniNNet.py nn_version 78 (:167-192,264-271; with gods, A = 1782) with the reference's parameter names:
    conv3x3(2->64), ten InvertedResidual blocks, meta_fc (gods plane -> 32 features), HeadWithMeta heads (bottlenecks 4 / 2)."""
    version = 78

    def __init__(self, num_players=2, action_size=1782, dropout=0.0):
        super().__init__()
        self.P, self.A, self.dropout = num_players, action_size, dropout
        self.first_layer = nn.Conv2d(2, 64, 3, padding=1, bias=False)
        self.trunk = nn.Sequential(*[_MBBlock2d(64, 192) for _ in range(10)])
        self.meta_fc = nn.Sequential(nn.Flatten(1), nn.Linear(25, 32), nn.ReLU())
        self.head_PI, self.head_V = _Head2d(64, 4, action_size, False, meta=32), _Head2d(64, 2, num_players, True, meta=32)
        self.register_buffer('lowvalue', torch.FloatTensor([-1e8]))

    def forward(self, boards, valid_actions):
        x = boards.reshape(-1, 5, 5, 3).float().permute(0, 3, 1, 2)
        f = self.trunk(self.first_layer(x[:, :2]))
        meta = self.meta_fc(x[:, 2:3])
        v = self.head_V(f, meta)
        pi = torch.where(valid_actions.bool(), self.head_PI(f, meta), self.lowvalue)
        return F.log_softmax(pi, dim=1), torch.tanh(v)


def loss_pi(target_pi, out_log_pi):                                            # GenericNNetWrapper.py:179-181
    return F.kl_div(out_log_pi, target_pi, reduction='batchmean')


def loss_v(target_z, target_q, out_v, q_weight):                               # :189-191
    t = (target_z + q_weight * target_q) / (1.0 + q_weight)
    return torch.sum((t - out_v) ** 2) / (target_z.shape[0] * target_z.shape[-1])


def train(module, examples, learn_rate=3e-3, batch_size=512, epochs=2, q_weight=0.5, device='cuda:0', seed=None, log=None, board_shape=None):
    """examples = (boards int8[n,S], pi f32[n,A], z f32[n,P], valids u8/bool[n,A], q f32[n,P]) tensors or arrays.
    board_shape: give it for a module that expects the reference's inputs (float boards of getBoardSize(), bool valids:
    GenericNNetWrapper.py:60-63) instead of the engine modules' flat int8 boards.  Returns the list of (pi loss, v loss) per step."""
    boards, pi, z, valids, q = [torch.as_tensor(np.asarray(x.cpu()) if hasattr(x, 'cpu') else x).to(device) for x in examples[:5]]
    n = boards.shape[0]
    valids = valids.bool()
    if board_shape is not None:
        boards = boards.reshape((n,) + tuple(board_shape)).to(torch.float32)
    steps_per_epoch = n // batch_size
    if steps_per_epoch == 0:
        raise ValueError('fewer examples (%d) than batch_size (%d)' % (n, batch_size))
    module.to(device).train()
    opt = torch.optim.AdamW(module.parameters(), lr=learn_rate)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=learn_rate, steps_per_epoch=steps_per_epoch, epochs=epochs)
    gen = torch.Generator(device='cpu')
    if seed is not None:
        gen.manual_seed(seed)
    hist = []
    for ep in range(epochs):
        for _ in range(steps_per_epoch):
            ids = torch.randperm(n, generator=gen)[:batch_size].to(device)   # np.random.choice(n, batch, replace=False) :57
            opt.zero_grad(set_to_none=True)
            out_pi, out_v = module(boards[ids], valids[ids])
            l_pi = loss_pi(pi[ids].float(), out_pi)
            l_v = loss_v(z[ids].float(), q[ids].float(), out_v, q_weight)
            (l_pi + 0.25 * l_v).backward()
            opt.step()
            sched.step()
            hist.append((l_pi.item(), l_v.item()))
        if log:
            log('epoch %d: pi loss %.4f  v loss %.4f' % (ep + 1, np.mean([h[0] for h in hist[-steps_per_epoch:]]),
                                                         np.mean([h[1] for h in hist[-steps_per_epoch:]])))
    module.eval()
    return hist
