"""NeuralNet.predict for the engine: one batched PyTorch-ROCm forward over the leaf batch (the only MFMA work on the
path).  Replaces GenericNNetWrapper.predict / predict_client / predict_server (GenericNNetWrapper.py:94-157): instead of
N threads time-slicing one core to build an ONNX batch of N, the forest's select kernel writes all T leaves and this
module evaluates them in one pass.

SplendorV80 re-expresses the reference's nn_version == 80 network (splendor/SplendorNNet.py:262-283,397-440 and the
blocks :148-202) in plain torch -- no torchvision -- in channels-last layout [B, 7, C] so every token-axis Linear is a
plain GEMM, with BatchNorm folded into the GEMM weights (eval mode) and the Flatten permutation folded into the first
head Linear.  It loads the reference's state_dict key names unchanged (checkpoint compatibility)."""
import numpy as np
import torch
import torch.nn.functional as F


def _fold_bn(sd, prefix, eps=1e-5):
    g, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    m, v = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    s = g / torch.sqrt(v + eps)
    return s, b - m * s


class _Block:
    """InvertedResidual1d (SplendorNNet.py:189-202) with folded BN; activations: ReLU or Hardswish; SE avg/max."""

    def __init__(self, sd, prefix, use_hs, setype):
        self.use_hs, self.setype = use_hs, setype
        s, b = _fold_bn(sd, prefix + '.expand.norm')
        self.We = (sd[prefix + '.expand.linear.weight'] * s[:, None]).t().contiguous()      # [Cin, Cexp]
        self.be = b
        s, b = _fold_bn(sd, prefix + '.depthwise.norm')
        self.Wd = sd[prefix + '.depthwise.linear.weight'].contiguous()                      # [7, 7] (out, in)
        self.sd, self.bd = s, b                                                             # per channel
        self.W1 = sd[prefix + '.se.fc1.weight'].t().contiguous()
        self.b1 = sd[prefix + '.se.fc1.bias']
        self.W2 = sd[prefix + '.se.fc2.weight'].t().contiguous()
        self.b2 = sd[prefix + '.se.fc2.bias']
        s, b = _fold_bn(sd, prefix + '.project.norm')
        self.Wp = (sd[prefix + '.project.linear.weight'] * s[:, None]).t().contiguous()     # [Cexp, Cout]
        self.bp = b

    def tensors(self):
        return ['We', 'be', 'Wd', 'sd', 'bd', 'W1', 'b1', 'W2', 'b2', 'Wp', 'bp']

    def act(self, x):
        return F.hardswish(x) if self.use_hs else F.relu(x)

    def __call__(self, x):                      # x: [B, 7, C]
        h = self.act(torch.matmul(x, self.We) + self.be)                       # expand  [B,7,Cexp]
        h = torch.matmul(self.Wd.to(h.dtype), h) * self.sd + self.bd           # depthwise Linear(7->7) over L, then BN
        h = self.act(h)
        pooled = h.mean(dim=1) if self.setype == 'avg' else h.amax(dim=1)      # SE squeeze over L  [B,Cexp]
        sc = F.hardsigmoid(torch.addmm(self.b2, F.relu(torch.addmm(self.b1, pooled, self.W1)), self.W2))
        h = h * sc[:, None, :]
        return torch.matmul(h, self.Wp) + self.bp + x                          # project + residual


class SplendorV80:
    """forward(board int8/float [B,56,7], valid bool [B,81]) -> (pi probabilities f32 [B,81], v f32 [B,P])."""

    def __init__(self, state_dict, num_players=2, device='cuda:0', dtype=torch.float32):
        sd = {k: torch.as_tensor(v).float() for k, v in state_dict.items()}
        self.P = num_players
        self.nb_vect = 32 + 10 * num_players + num_players * num_players
        self.A = 81
        s, b = _fold_bn(sd, 'first_layer.norm')
        self.W0 = (sd['first_layer.linear.weight'] * s[:, None]).t().contiguous()
        self.b0 = b
        self.trunk = _Block(sd, 'trunk.0', False, 'avg')
        self.head_pi = _Block(sd, 'output_layers_PI.0', True, 'max')
        self.head_v = _Block(sd, 'output_layers_V.0', True, 'max')
        C = self.nb_vect

        def perm(w):   # reference flattens [B, C, 7] (index c*7+l); ours is [B, 7, C] (index l*C+c)
            return w.view(w.shape[0], C, 7).permute(0, 2, 1).reshape(w.shape[0], 7 * C).t().contiguous()
        self.Wpi1, self.bpi1 = perm(sd['output_layers_PI.2.weight']), sd['output_layers_PI.2.bias']
        self.Wpi2, self.bpi2 = sd['output_layers_PI.4.weight'].t().contiguous(), sd['output_layers_PI.4.bias']
        self.Wv1, self.bv1 = perm(sd['output_layers_V.2.weight']), sd['output_layers_V.2.bias']
        self.Wv2, self.bv2 = sd['output_layers_V.4.weight'].t().contiguous(), sd['output_layers_V.4.bias']
        self.to(device, dtype)

    def to(self, device, dtype=torch.float32):
        self.device, self.dtype = torch.device(device), dtype
        for name in ['W0', 'b0', 'Wpi1', 'bpi1', 'Wpi2', 'bpi2', 'Wv1', 'bv1', 'Wv2', 'bv2']:
            setattr(self, name, getattr(self, name).to(self.device, dtype))
        for blk in (self.trunk, self.head_pi, self.head_v):
            for name in blk.tensors():
                setattr(blk, name, getattr(blk, name).to(self.device, dtype))
        return self

    @staticmethod
    def from_npz(path, **kw):
        z = np.load(path)
        sd = {k[3:]: z[k] for k in z.files if k.startswith('sd/')}
        return SplendorV80(sd, **kw)

    @staticmethod
    def random_init(num_players=2, seed=0, **kw):
        """random weights of the V80 architecture (for runs without a checkpoint)"""
        g = torch.Generator().manual_seed(seed)
        C, E, Q = 32 + 10 * num_players + num_players * num_players, 0, 0
        E = 3 * C
        Q = max(8, int(E // 4 + 4) // 8 * 8)
        sd = {}

        def lin(name, o, i, bias=True):
            sd[name + '.weight'] = (torch.rand(o, i, generator=g) * 2 - 1) * (6.0 / i) ** 0.5
            if bias:
                sd[name + '.bias'] = torch.zeros(o)

        def bn(name, c):
            sd[name + '.weight'], sd[name + '.bias'] = torch.ones(c), torch.zeros(c)
            sd[name + '.running_mean'], sd[name + '.running_var'] = torch.zeros(c), torch.ones(c)

        def block(p):
            lin(p + '.expand.linear', E, C, False); bn(p + '.expand.norm', E)
            lin(p + '.depthwise.linear', 7, 7, False); bn(p + '.depthwise.norm', E)
            lin(p + '.se.fc1', Q, E); lin(p + '.se.fc2', E, Q)
            lin(p + '.project.linear', C, E, False); bn(p + '.project.norm', C)
        lin('first_layer.linear', C, C, False); bn('first_layer.norm', C)
        block('trunk.0'); block('output_layers_PI.0'); block('output_layers_V.0')
        lin('output_layers_PI.2', 81, 7 * C); lin('output_layers_PI.4', 81, 81)
        lin('output_layers_V.2', num_players, 7 * C); lin('output_layers_V.4', num_players, num_players)
        return SplendorV80(sd, num_players=num_players, **kw)

    @torch.no_grad()
    def forward(self, boards, valids):
        B = boards.shape[0]
        x = boards.reshape(B, self.nb_vect, 7).to(self.dtype).transpose(1, 2)                 # [B,7,C] view
        x = torch.matmul(x, self.W0) + self.b0                                                 # first_layer (+BN)
        x = self.trunk(x)
        hp = self.head_pi(x).reshape(B, -1)
        logits = torch.addmm(self.bpi2, F.relu(torch.addmm(self.bpi1, hp, self.Wpi1)), self.Wpi2).float()
        hv = self.head_v(x).reshape(B, -1)
        v = torch.tanh(torch.addmm(self.bv2, F.relu(torch.addmm(self.bv1, hv, self.Wv1)), self.Wv2).float())
        logits = torch.where(valids.bool(), logits, torch.full_like(logits, -1e8))            # SplendorNNet.py:404
        pi = torch.softmax(logits, dim=1)              # exp(log_softmax) of GenericNNetWrapper.py:107,119
        return pi.contiguous(), v.contiguous()

    # NeuralNet.predict-compatible entry points
    def predict_batch(self, boards, valids):
        return self.forward(boards, valids)

    def predict(self, board, valid_actions):
        b = torch.from_numpy(np.ascontiguousarray(board, dtype=np.int8))[None].to(self.device)
        va = torch.from_numpy(np.asarray(valid_actions).astype(np.bool_))[None].to(self.device)
        pi, v = self.forward(b, va)
        return pi[0].cpu().numpy(), v[0].cpu().numpy()
