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

    def __call__(self, x):                      # x: [B, L, C]
        h = self.act(torch.matmul(x, self.We) + self.be)                       # expand  [B,7,Cexp]
        h = torch.matmul(self.Wd.to(h.dtype), h) * self.sd + self.bd           # depthwise Linear(7->7) over L, then BN
        h = self.act(h)
        pooled = h.mean(dim=1) if self.setype == 'avg' else h.amax(dim=1)      # SE squeeze over L  [B,Cexp]
        sc = F.hardsigmoid(torch.addmm(self.b2, F.relu(torch.addmm(self.b1, pooled, self.W1)), self.W2))
        h = h * sc[:, None, :]
        out = torch.matmul(h, self.Wp) + self.bp                               # project (+BN)
        return out + x if self.Wp.shape[1] == x.shape[-1] else out             # use_res_connect: in == out channels


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

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        return cls({k[3:]: z[k] for k in z.files if k.startswith('sd/')}, **kw)

    @classmethod
    def random_init(cls, num_players=2, seed=0, **kw):
        """random weights of the V80 architecture (for runs without a checkpoint)"""
        return cls(cls.random_state_dict(num_players, seed), num_players=num_players, **kw)

    @staticmethod
    def random_state_dict(num_players=2, seed=0):
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
        return sd

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


class SplendorV80Hip(SplendorV80):
    """Same network, same weights, evaluated by the engine's own gfx950 kernels (azg_nn_* in include/azg.h) instead of
    ~70 torch ops: 9 skinny fp32 MFMA GEMMs (k_linear, with bias / activation / residual / SE-scale fused), 3
    depthwise+BN+act+pool kernels, 3 SE kernels, one layout kernel and one softmax/value kernel per leaf batch."""

    def __init__(self, state_dict, num_players=2, device='cuda:0', max_batch=4096, split=True, h2=None):
        """h2 (default for the 2-player geometry): the one-launch forward on fp16 hi+lo split operands with token-major tiles
        (azg_nn_v80_forward_h2, csrc/nn_v80_h2.hip.h; same 1e-5 contract).  Otherwise split: the tile the blocks read is kept as
        three bf16 planes and the expand GEMMs run on bf16 x 3 operands (azg_nn_v80_forward_split); False = f32 MFMAs throughout"""
        super().__init__(state_dict, num_players=num_players, device=device, dtype=torch.float32)
        from . import _lib
        self._lib = _lib
        self.split = bool(split)
        self.h2 = (self.nb_vect == 56) if h2 is None else bool(h2)
        self.C = self.nb_vect
        self.E = 3 * self.C
        self.Q = self.trunk.W1.shape[1]
        self.weight_stationary = True
        self.fused_blocks = True
        self.fused_net = True          # whole forward in one launch (azg_nn_v80_forward)
        self._bias_pad = {}
        self._prepare()
        self._alloc(max_batch)

    def _alloc(self, B):
        d, f = self.device, torch.float32
        self.maxB = B
        self.x0 = torch.empty((B * 7, self.C), dtype=f, device=d)
        self.x1 = torch.empty((B * 7, self.C), dtype=f, device=d)
        self.x2 = torch.empty((B * 7, self.C), dtype=f, device=d)
        self.xh = torch.empty((B * 7, self.C), dtype=f, device=d)
        self.h = torch.empty((B * 7, self.E), dtype=f, device=d)
        self.pooled = torch.empty((B, self.E), dtype=f, device=d)
        self.sc = torch.empty((B, self.E), dtype=f, device=d)
        self.se_h = torch.zeros((B, 48), dtype=f, device=d)        # Q = 40 used, zero-padded to a multiple of 16 for K
        self.hid_pi = torch.zeros((B, 96), dtype=f, device=d)      # 81 used, zero-padded to a multiple of 4 for K
        self.logits = torch.empty((B, 96), dtype=f, device=d)
        self.hid_v = torch.zeros((B, 16), dtype=f, device=d)
        self.pi = torch.empty((B, self.A), dtype=f, device=d)
        self.v = torch.empty((B, self.P), dtype=f, device=d)

    def clone_buffers(self):
        """a second evaluator sharing the (read-only) weights but with its own activation buffers (concurrent streams)"""
        import copy
        other = copy.copy(self)
        other._alloc(self.maxB)
        return other

    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _pad_w(W):
        """[K][N] -> zero-padded [Kp][NP], Kp multiple of 16, NP/16 in {1,4,6,11} (k_linear's LDS / fragment layout)"""
        K, N = W.shape
        Kp = (K + 15) // 16 * 16
        nt = (N + 15) // 16
        out = torch.zeros((Kp, nt * 16), dtype=torch.float32, device=W.device)
        out[:K, :N] = W
        return out.contiguous()

    def _prepare(self):
        self.pW0 = self._pad_w(self.W0)
        for blk in (self.trunk, self.head_pi, self.head_v):
            blk.pWe, blk.pWp = self._pad_w(blk.We), self._pad_w(blk.Wp)
            blk.pW1, blk.pW2 = self._pad_w(blk.W1), self._pad_w(blk.W2)
        self.pWpi1, self.pWpi2 = self._pad_w(self.Wpi1), self._pad_w(self.Wpi2)
        self.pWv1 = self._pad_w(self.Wv1)
        for blk in (self.trunk, self.head_pi, self.head_v):
            self._block_ptrs(blk)
        self._net_ptrs()

    def _net_ptrs(self):
        """device pointer table of azg_nn_v80_forward (include/azg.h): first layer, 3 blocks, head Linears re-indexed to the
        in-LDS flatten k = l*60 + c"""
        import ctypes as C
        assert (self.trunk.use_hs, self.trunk.setype) == (False, 'avg')
        assert (self.head_pi.use_hs, self.head_pi.setype) == (True, 'max') and (self.head_v.use_hs, self.head_v.setype) == (True, 'max')
        assert self.C == 56 and self.A == 81
        d, f = self.device, torch.float32

        def pad(t, shape):
            out = torch.zeros(shape, dtype=f, device=d)
            out[tuple(slice(0, n) for n in t.shape)] = t
            return out.contiguous()

        def flat60(Wf, ncols):        # [7*56][N] (k = l*56 + c) -> [432][ncols] (k = l*60 + c)
            N = Wf.shape[1]
            out = torch.zeros((432, ncols), dtype=f, device=d)
            out[:420].view(7, 60, ncols)[:, :56, :N] = Wf.view(7, 56, N)
            return out.contiguous()
        head = [self._frag(flat60(self.Wpi1, 96)), pad(self.bpi1, (96,)), self._frag(pad(self.Wpi2, (96, 96))),
                pad(self.bpi2, (96,)), self._frag(flat60(self.Wv1, 16)), pad(self.bv1, (16,)), self.Wv2.contiguous(),
                self.bv2.contiguous()]
        first = [self._frag(pad(self.W0, (64, 64)), True), pad(self.b0, (64,))]
        self._net_keep = first + self.trunk._keep + self.head_pi._keep + self.head_v._keep + head
        assert len(self._net_keep) == 43
        self.net_ptrs = (C.c_void_p * 43)(*[t.data_ptr() for t in self._net_keep])

        def split_frag(Wp):                # zero-padded [K][N], K % 32 == 0 -> [N/16 tiles][K/32 chunks][3 planes hi, mid, lo][64 lanes][8] bf16
            K, N = Wp.shape
            m = Wp.contiguous().float()
            hi = m.to(torch.bfloat16)
            r1 = m - hi.float()
            mid = r1.to(torch.bfloat16)
            lo = (r1 - mid.float()).to(torch.bfloat16)
            pl = torch.stack([hi, mid, lo]).view(3, K // 32, 4, 8, N // 16, 16)           # plane, chunk, g, j, tile, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                     # tile, chunk, plane, g, r, j
        keep = list(self._net_keep)
        for bi, blk in enumerate((self.trunk, self.head_pi, self.head_v)):
            keep[2 + 11 * bi] = split_frag(blk.pWe)              # [64][176] (K 56 -> 64 zero padded)
        self._net_keep_split = keep
        self.net_ptrs_split = (C.c_void_p * 43)(*[t.data_ptr() for t in keep])
        self._h2_ptrs()

    def _h2_ptrs(self):
        """pointer table + descale factors of azg_nn_v80_forward_h2 (include/azg.h): every matrix zero padded to K % 32 == 0,
        N % 16 == 0, scaled by 2^k (max |w| * 2^k in (2^11, 2^12]) and split into f16 hi / lo fragments"""
        import ctypes as C
        import math
        d, f = self.device, torch.float32

        def pad(t, shape):
            out = torch.zeros(shape, dtype=f, device=d)
            out[tuple(slice(0, n) for n in t.shape)] = t
            return out.contiguous()

        def frag(W, K, N):
            m = pad(W.to(f), (K, N))
            k = 12 - int(math.ceil(math.log2(max(float(m.abs().max()), 1e-30))))      # (an all-zero matrix: any scale)
            m = m * (2.0 ** k)
            hi = m.to(torch.float16)
            lo = (m - hi.float()).to(torch.float16)
            assert bool(torch.isfinite(hi.float()).all())
            pl = torch.stack([hi, lo]).view(2, K // 32, 4, 8, N // 16, 16)                 # plane, chunk, g, j, tile, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1), (2.0 ** -k) / 64.0   # tile, chunk, plane, g, r, j

        def flat64(Wf, N):            # [7*56][N] (k = l*56 + c) -> [448][N] (k = l*64 + c)
            out = torch.zeros((448, Wf.shape[1]), dtype=f, device=d)
            out.view(7, 64, Wf.shape[1])[:, :56, :] = Wf.view(7, 56, Wf.shape[1])
            return out
        keep, desc = [], []
        t, s = frag(self.W0, 64, 64)
        keep += [t, pad(self.b0, (64,))]
        desc.append(s)
        for blk in (self.trunk, self.head_pi, self.head_v):
            we, se = frag(blk.We, 64, 176)
            w1, s1 = frag(blk.W1, 192, 48)
            w2, s2 = frag(blk.W2, 64, 176)
            wp, sp = frag(blk.Wp, 192, 64)
            keep += [we, pad(blk.be, (176,)), (blk.Wd / 6.0 if blk.use_hs else blk.Wd).contiguous().to(f), pad(blk.sd, (176,)), pad(blk.bd, (176,)), w1, pad(blk.b1, (48,)),
                     w2, pad(blk.b2, (176,)), wp, pad(blk.bp, (64,))]
            desc += [se, s1, s2, sp]
        wpi1, spi1 = frag(flat64(self.Wpi1, 96), 448, 96)
        wpi2, spi2 = frag(self.Wpi2, 96, 96)
        wv1, sv1 = frag(flat64(self.Wv1, 16), 448, 16)
        keep += [wpi1, pad(self.bpi1, (96,)), wpi2, pad(self.bpi2, (96,)), wv1, pad(self.bv1, (16,)), self.Wv2.contiguous(), self.bv2.contiguous()]
        desc += [spi1, spi2, sv1]
        assert len(keep) == 43 and len(desc) == 16
        self._net_keep_h2 = keep
        self.net_ptrs_h2 = (C.c_void_p * 43)(*[t.data_ptr() for t in keep])
        self.descale_h2 = (C.c_float * 16)(*desc)

    def _linear(self, A, lda, Wp, bias, out, ldc, M, K, N, act=0, R=None, ldr=0, rowscale=None, rpg=0, ksplit=0):
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        if self.weight_stationary:
            bp = None
            if bias is not None:
                key = bias.data_ptr()
                bp = self._bias_pad.get(key)
                if bp is None:
                    bp = torch.zeros(Wp.shape[1], dtype=torch.float32, device=bias.device)
                    bp[:bias.numel()] = bias
                    self._bias_pad[key] = bp
            self._lib.check(self._lib.lib().azg_nn_linear_ws(p(A), lda, p(Wp), Wp.shape[0], Wp.shape[1], p(bp), p(R), ldr,
                                                             p(rowscale), rpg, p(out), ldc, M, K, N, act, self._stream()))
            return
        self._lib.check(self._lib.lib().azg_nn_linear(p(A), lda, p(Wp), Wp.shape[0], Wp.shape[1], p(bias), p(R), ldr,
                                                      p(rowscale), rpg, p(out), ldc, M, K, N, act, ksplit,
                                                      self._stream()))

    @staticmethod
    def _frag(Wp, half_last=False):
        """zero-padded [Kp][NP] -> MFMA fragment order [NP/16][Kp/16][64 lanes][4]:
        frag[nt][c][lane][j] = Wp[16c + 4*(lane>>4) + j][16nt + (lane&15)]  (FRAG in csrc/nn_kernels.hip.h).
        half_last: the last chunk holds only 8 rows of K; lane group g gets rows 16c + 2g + {0, 1} in j = 0, 1"""
        Kp, NP = Wp.shape
        assert Kp % 16 == 0 and NP % 16 == 0
        if half_last:
            last = Wp[Kp - 16:].clone()
            assert float(last[8:].abs().max()) == 0.0
            Wp = Wp.clone()
            Wp[Kp - 16:] = 0
            for g in range(4):
                Wp[Kp - 16 + 4 * g: Kp - 16 + 4 * g + 2] = last[2 * g: 2 * g + 2]
        return Wp.view(Kp // 16, 4, 4, NP // 16, 16).permute(3, 0, 1, 4, 2).contiguous().view(-1)

    def _block_ptrs(self, blk):
        import ctypes as C

        def pad1(v, n):
            out = torch.zeros(n, dtype=torch.float32, device=v.device)
            out[:v.numel()] = v
            return out
        blk._keep = [self._frag(blk.pWe, True), pad1(blk.be, 176), blk.Wd.contiguous(), blk.sd.contiguous(), blk.bd.contiguous(),
                     self._frag(blk.pW1, True), pad1(blk.b1, 48), self._frag(blk.pW2), pad1(blk.b2, 176), self._frag(blk.pWp, True),
                     pad1(blk.bp, 64)]
        assert tuple(blk.pWe.shape) == (64, 176) and tuple(blk.pW1.shape) == (176, 48)
        assert tuple(blk.pW2.shape) == (48, 176) and tuple(blk.pWp.shape) == (176, 64)
        blk.ptrs = (C.c_void_p * 11)(*[t.data_ptr() for t in blk._keep])

    def _block(self, blk, xin, xout, B):
        import ctypes as C
        L = self._lib.lib()
        if self.fused_blocks:
            self._lib.check(L.azg_nn_v80_block(C.c_void_p(xin.data_ptr()), C.c_void_p(xout.data_ptr()), blk.ptrs, B,
                                               2 if blk.use_hs else 1, 0 if blk.setype == 'avg' else 1, self._stream()))
            return
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        act = 2 if blk.use_hs else 1
        M = B * 7
        self._linear(xin, self.C, blk.pWe, blk.be, self.h, self.E, M, self.C, self.E, act=act)           # expand+BN+act
        self._lib.check(L.azg_nn_dw_pool(p(self.h), self.E, p(blk.Wd), p(blk.sd), p(blk.bd), p(self.pooled), B, self.E,
                                         act, 0 if blk.setype == 'avg' else 1, self._stream()))           # depthwise+BN+act+squeeze
        self._linear(self.pooled, self.E, blk.pW1, blk.b1, self.se_h, 48, B, self.E, self.Q, act=1, ksplit=1)   # SE fc1+ReLU
        self._linear(self.se_h, 48, blk.pW2, blk.b2, self.sc, self.E, B, 48, self.E, act=3, ksplit=1)            # SE fc2+Hardsigmoid
        self._linear(self.h, self.E, blk.pWp, blk.bp, xout, self.C, M, self.E, self.C, act=0, R=xin, ldr=self.C,
                     rowscale=self.sc, rpg=7)                                                             # SE*h @ Wp + BN + residual

    @torch.no_grad()
    def forward(self, boards, valids):
        import ctypes as C
        B = boards.shape[0]
        if B > self.maxB:
            self._alloc(B)
        L = self._lib.lib()
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        boards = boards.reshape(B, -1)
        assert boards.dtype == torch.int8 and boards.is_contiguous() and boards.is_cuda
        valids = valids if valids.dtype == torch.uint8 else valids.to(torch.uint8)
        if self.fused_net and self.h2:
            self._lib.check(L.azg_nn_v80_forward_h2(p(boards), p(valids), self.net_ptrs_h2, self.descale_h2, B, self.P, p(self.pi), p(self.v),
                                                    self._stream()))
            return self.pi[:B], self.v[:B]
        if self.fused_net:
            fwd, ptrs = (L.azg_nn_v80_forward_split, self.net_ptrs_split) if self.split else (L.azg_nn_v80_forward, self.net_ptrs)
            self._lib.check(fwd(p(boards), p(valids), ptrs, B, self.P, p(self.x2), p(self.pi), p(self.v), self._stream()))
            return self.pi[:B], self.v[:B]
        self._lib.check(L.azg_nn_board_to_x(p(boards), p(self.x0), B, self.C, self._stream()))
        self._linear(self.x0, self.C, self.pW0, self.b0, self.x1, self.C, B * 7, self.C, self.C)           # first_layer
        self._block(self.trunk, self.x1, self.x2, B)
        self._block(self.head_pi, self.x2, self.xh, B)
        self._linear(self.xh, 7 * self.C, self.pWpi1, self.bpi1, self.hid_pi, 96, B, 7 * self.C, self.A, act=1, ksplit=1)
        self._linear(self.hid_pi, 96, self.pWpi2, self.bpi2, self.logits, 96, B, 96, self.A, ksplit=1)
        self._block(self.head_v, self.x2, self.xh, B)
        self._linear(self.xh, 7 * self.C, self.pWv1, self.bv1, self.hid_v, 16, B, 7 * self.C, self.P, ksplit=1)
        self._lib.check(L.azg_nn_heads_out(p(self.logits), 96, p(valids), p(self.hid_v), 16, p(self.Wv2), p(self.bv2),
                                           p(self.pi), p(self.v), B, self.A, self.P, self._stream()))
        return self.pi[:B], self.v[:B]



class AzulV84(SplendorV80):
    """azul/AzulNNet.py nn_version == 84 (:91-113,130-142): same building blocks on a [B, 23, 6] board -- trunk block
    23->115->23, policy head block 23->115->46 (no residual) + Linear(276,180)+ReLU+Linear, value head block 23->46->23."""

    def __init__(self, state_dict, num_players=2, device='cuda:0', dtype=torch.float32):
        sd = {k: torch.as_tensor(v).float() for k, v in state_dict.items()}
        self.P = num_players
        self.nb_vect, self.L, self.A = 23, 6, 180
        s, b = _fold_bn(sd, 'first_layer.norm')
        self.W0 = (sd['first_layer.linear.weight'] * s[:, None]).t().contiguous()
        self.b0 = b
        self.trunk = _Block(sd, 'trunk.0', False, 'avg')
        self.head_pi = _Block(sd, 'output_layers_PI.0', True, 'avg')
        self.head_v = _Block(sd, 'output_layers_V.0', True, 'avg')
        L = self.L

        def perm(w, C):   # reference flattens [B, C, L] (index c*L+l); ours is [B, L, C] (index l*C+c)
            return w.view(w.shape[0], C, L).permute(0, 2, 1).reshape(w.shape[0], L * C).t().contiguous()
        self.Wpi1, self.bpi1 = perm(sd['output_layers_PI.2.weight'], 46), sd['output_layers_PI.2.bias']
        self.Wpi2, self.bpi2 = sd['output_layers_PI.4.weight'].t().contiguous(), sd['output_layers_PI.4.bias']
        self.Wv1, self.bv1 = perm(sd['output_layers_V.2.weight'], 23), sd['output_layers_V.2.bias']
        self.Wv2, self.bv2 = sd['output_layers_V.4.weight'].t().contiguous(), sd['output_layers_V.4.bias']
        self.to(device, dtype)

    @torch.no_grad()
    def forward(self, boards, valids):
        B = boards.shape[0]
        x = boards.reshape(B, self.nb_vect, self.L).to(self.dtype).transpose(1, 2)
        x = torch.matmul(x, self.W0) + self.b0
        x = self.trunk(x)
        hp = self.head_pi(x).reshape(B, -1)
        logits = torch.addmm(self.bpi2, F.relu(torch.addmm(self.bpi1, hp, self.Wpi1)), self.Wpi2).float()
        hv = self.head_v(x).reshape(B, -1)
        v = torch.tanh(torch.addmm(self.bv2, F.relu(torch.addmm(self.bv1, hv, self.Wv1)), self.Wv2).float())
        logits = torch.where(valids.bool(), logits, torch.full_like(logits, -1e8))
        return torch.softmax(logits, dim=1).contiguous(), v.contiguous()


class MobileNet1d(AzulV84):
    """Any net of the reference's one-trunk-block MobileNetV3-1d family, geometry read off the state_dict: first_layer, ONE trunk block
    (ReLU, mean squeeze), one policy-head and one value-head block (Hardswish, `head_se` squeeze), Flatten + Linear + ReLU + Linear heads.
    The shipped nets of two more games are of this shape (their checkpoints load unchanged):
      minivilles/MinivillesNNet.py:101-123 nn_version 82 -- [B, 58, 2] board (2 players), blocks 58 -> 174 -> 58, heads Linear(116, 21) / (116, 2)
      thelittleprince/TLPNNet.py:175-196 nn_version 83   -- [B, 55, 15] board (3 players), blocks 55 -> 82 -> 55, heads Linear(825, 9) / (825, 3)
    (TLP nn_version 80 / 82 differ in the expansion factor only and load the same way.)"""

    def __init__(self, state_dict, num_players=None, head_se='max', device='cuda:0', dtype=torch.float32):
        sd = {k: torch.as_tensor(v).float() for k, v in state_dict.items()}
        self.nb_vect = int(sd['first_layer.linear.weight'].shape[0])
        self.L = L = int(sd['trunk.0.depthwise.linear.weight'].shape[0])
        self.A = int(sd['output_layers_PI.4.weight'].shape[0])
        self.P = int(sd['output_layers_V.4.weight'].shape[0])
        assert num_players in (None, self.P)
        s, b = _fold_bn(sd, 'first_layer.norm')
        self.W0 = (sd['first_layer.linear.weight'] * s[:, None]).t().contiguous()
        self.b0 = b
        self.trunk = _Block(sd, 'trunk.0', False, 'avg')
        self.head_pi = _Block(sd, 'output_layers_PI.0', True, head_se)
        self.head_v = _Block(sd, 'output_layers_V.0', True, head_se)

        def perm(w, C):   # reference flattens [B, C, L] (index c*L+l); ours is [B, L, C] (index l*C+c)
            return w.view(w.shape[0], C, L).permute(0, 2, 1).reshape(w.shape[0], L * C).t().contiguous()
        self.Wpi1, self.bpi1 = perm(sd['output_layers_PI.2.weight'], self.head_pi.Wp.shape[1]), sd['output_layers_PI.2.bias']
        self.Wpi2, self.bpi2 = sd['output_layers_PI.4.weight'].t().contiguous(), sd['output_layers_PI.4.bias']
        self.Wv1, self.bv1 = perm(sd['output_layers_V.2.weight'], self.head_v.Wp.shape[1]), sd['output_layers_V.2.bias']
        self.Wv2, self.bv2 = sd['output_layers_V.4.weight'].t().contiguous(), sd['output_layers_V.4.bias']
        self.to(device, dtype)


class MinivillesV82(MobileNet1d):
    """minivilles/MinivillesNNet.py nn_version == 82 (:101-123,166-172), the net of minivilles/pretrained_2players.pt"""


class TLPV83(MobileNet1d):
    """thelittleprince/TLPNNet.py nn_version == 83 (:175-196,211-217), the net of thelittleprince/pretrained_3players.pt"""


class MobileNet1dHip:
    """The MobileNetV3-1d policy/value nets of any geometry (Splendor V80 for 2-4 players: C = 32 + 10n + n^2 channels x 7
    tokens; Azul V84: 23 channels x 6 tokens, AzulNNet.py:91-113) evaluated by the engine's gfx950 kernels instead of ~65
    torch ops per leaf batch: azg_nn_board_to_x_ld, then per InvertedResidual1d block (SplendorNNet.py:189-202) expand GEMM
    (bias + activation fused), depthwise+BN+act+squeeze, the two SE GEMMs and the project GEMM (SE scale on the operand,
    bias, residual fused); flatten->Linear heads through the K-split GEMM; masked softmax / tanh tail.  17 launches.
    Wraps a SplendorV80 / AzulV84 instance (folded-BN fp32 weights); channel counts are zero-padded to multiples of 4
    (row strides) and of 16 (weight tiles), which leaves the results unchanged."""

    _WS_CHUNKS = (1, 2, 3, 4, 6, 8, 11, 17, 25)

    def __init__(self, base, max_batch=4096, fused=True, h2=True):
        from . import _lib
        self._lib = _lib
        self.base = base
        self.h2 = h2
        self.device = base.device
        self.P, self.A, self.C = base.P, base.A, base.nb_vect
        self.L = getattr(base, 'L', 7)
        self.nb_vect = base.nb_vect
        assert base.dtype == torch.float32 and self.device.type == 'cuda'
        r4, r16 = (lambda n: (n + 3) // 4 * 4), (lambda n: (n + 15) // 16 * 16)
        self.Cp = r4(self.C)
        d = self.device

        def padw(W, Kp, NP):
            out = torch.zeros((Kp, NP), dtype=torch.float32, device=d)
            out[:W.shape[0], :W.shape[1]] = W
            return out

        def padv(v, n):
            out = torch.zeros(n, dtype=torch.float32, device=d)
            out[:v.numel()] = v
            return out
        self.pW0, self.pb0 = padw(base.W0, r16(self.Cp), r16(self.C)), padv(base.b0, r16(self.C))
        self.blocks = []
        for blk in (base.trunk, base.head_pi, base.head_v):
            cin, E = blk.We.shape
            Q, cout = blk.W1.shape[1], blk.Wp.shape[1]
            g = dict(cin=cin, cinp=r4(cin), E=E, Ep=r4(E), Q=Q, Qp=r16(Q), cout=cout, coutp=r4(cout), act=2 if blk.use_hs else 1,
                     pool_max=0 if blk.setype == 'avg' else 1, res=cin == cout)
            g['We'], g['be'] = padw(blk.We, r16(g['cinp']), r16(g['Ep'])), padv(blk.be, r16(g['Ep']))
            g['Wd'], g['sd'], g['bd'] = blk.Wd.contiguous(), padv(blk.sd, g['Ep']), padv(blk.bd, g['Ep'])
            g['W1'], g['b1'] = padw(blk.W1, r16(g['Ep']), g['Qp']), padv(blk.b1, g['Qp'])
            g['W2'], g['b2'] = padw(blk.W2, g['Qp'], r16(g['Ep'])), padv(blk.b2, r16(g['Ep']))
            g['Wp'], g['bp'] = padw(blk.Wp, r16(g['Ep']), r16(cout)), padv(blk.bp, r16(cout))
            self.blocks.append(g)
        L = self.L

        def flat(W, cout, coutp):          # rows l*cout + c  ->  l*coutp + c (pad rows zero), then tile padding
            N = W.shape[1]
            w = torch.zeros((L, coutp, N), dtype=torch.float32, device=d)
            w[:, :cout] = W.view(L, cout, N)
            return padw(w.view(L * coutp, N), r16(L * coutp), r16(N))
        gp, gv = self.blocks[1], self.blocks[2]
        self.Ap = r16(self.A)
        self.pWpi1, self.pWpi2 = flat(base.Wpi1, gp['cout'], gp['coutp']), padw(base.Wpi2, self.Ap, self.Ap)
        self.pWv1 = flat(base.Wv1, gv['cout'], gv['coutp'])
        self.bpi1, self.bpi2, self.bv1 = base.bpi1.contiguous(), base.bpi2.contiguous(), base.bv1.contiguous()
        self.Wv2, self.bv2 = base.Wv2.contiguous(), base.bv2.contiguous()
        self.geometry = {(7, 56): 0, (7, 71): 1, (7, 88): 2, (6, 23): 3, (2, 58): 4, (15, 55): 5}.get((self.L, self.C))    # AZG_NET_* of azg.h
        self.fused = fused and self.geometry is not None
        if not self.fused and self.geometry not in (0, 1, 2, 3):
            # the launch-per-layer path (azg_nn_linear's tile shapes) exists for the Splendor and Azul geometries only
            raise ValueError('MobileNet1dHip: geometry L=%d C=%d has the one-launch kernels only (fused=True)' % (self.L, self.C))
        if self.fused:
            self._pack_fused(padw, padv, r16)
        self._alloc(max_batch)

    def _pack_fused(self, padw, padv, r16):
        """the 43 weight pointers of azg_nn_mb1d_forward: matrices zero-padded to multiples of 16 and stored in MFMA
        fragment order, vectors zero-padded to multiples of 16"""
        import ctypes as C
        base, L, frag = self.base, self.L, SplendorV80Hip._frag
        fw = lambda W: frag(padw(W, r16(W.shape[0]), r16(W.shape[1])))  # noqa: E731
        keep = [fw(base.W0), padv(base.b0, r16(self.C))]
        for blk in (base.trunk, base.head_pi, base.head_v):
            E, Q, co = blk.We.shape[1], blk.W1.shape[1], blk.Wp.shape[1]
            keep += [fw(blk.We), padv(blk.be, r16(E)), blk.Wd.contiguous(), padv(blk.sd, r16(E)), padv(blk.bd, r16(E)),
                     fw(blk.W1), padv(blk.b1, r16(Q)), fw(blk.W2), padv(blk.b2, r16(E)), fw(blk.Wp), padv(blk.bp, r16(co))]
        co_pi, co_v = base.head_pi.Wp.shape[1], base.head_v.Wp.shape[1]
        OS = r16(max(self.C, co_pi)) + 4

        def flat(W, cout):                 # rows l*cout + c -> l*OS + c
            w = torch.zeros((L, OS, W.shape[1]), dtype=torch.float32, device=W.device)
            w[:, :cout] = W.view(L, cout, W.shape[1])
            return fw(w.view(L * OS, W.shape[1]))
        keep += [flat(base.Wpi1, co_pi), padv(base.bpi1, r16(self.A)), fw(base.Wpi2), padv(base.bpi2, r16(self.A)),
                 flat(base.Wv1, co_v), padv(base.bv1, 16), base.Wv2.contiguous(), base.bv2.contiguous()]
        assert len(keep) == 43
        self._fused_keep = keep
        self.fused_ptrs = (C.c_void_p * 43)(*[t.data_ptr() for t in keep])
        # azg_nn_mb1d_forward_h2: the same table with the matrices as f16 x 2 fragments (K padded to multiples of 32)
        import math
        r32 = lambda n: (r16(n) + 31) // 32 * 32  # noqa: E731
        desc = []

        def fh(W):
            m = padw(W, r32(W.shape[0]), r16(W.shape[1]))
            K, N = m.shape
            k = 12 - int(math.ceil(math.log2(max(float(m.abs().max()), 1e-30))))
            m = m * (2.0 ** k)
            hi = m.to(torch.float16)
            lo = (m - hi.float()).to(torch.float16)
            assert bool(torch.isfinite(hi.float()).all())
            desc.append((2.0 ** -k) / 64.0)
            pl = torch.stack([hi, lo]).view(2, K // 32, 4, 8, N // 16, 16)                # plane, chunk, g, j, tile, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                      # tile, chunk, plane, g, r, j

        def flat_rows(W, cout):
            w = torch.zeros((L, OS, W.shape[1]), dtype=torch.float32, device=W.device)
            w[:, :cout] = W.view(L, cout, W.shape[1])
            return w.view(L * OS, W.shape[1])
        k2 = list(keep)
        k2[0] = fh(base.W0)
        for b, blk in enumerate((base.trunk, base.head_pi, base.head_v)):
            o = 2 + 11 * b
            k2[o], k2[o + 5], k2[o + 7], k2[o + 9] = fh(blk.We), fh(blk.W1), fh(blk.W2), fh(blk.Wp)
        k2[35], k2[37], k2[39] = fh(flat_rows(base.Wpi1, co_pi)), fh(base.Wpi2), fh(flat_rows(base.Wv1, co_v))
        assert len(desc) == 16
        self._fused_keep_h2 = k2
        self.fused_ptrs_h2 = (C.c_void_p * 43)(*[t.data_ptr() for t in k2])
        self.descale_h2 = (C.c_float * 16)(*desc)

    def _alloc(self, B):
        d, f = self.device, torch.float32
        self.maxB = B
        M = B * self.L
        z = lambda *shape: torch.zeros(shape, dtype=f, device=d)  # noqa: E731   (pad columns must stay zero)
        self.x0, self.x1, self.x2 = z(M, self.Cp), z(M, self.Cp), z(M, self.blocks[0]['coutp'])
        Epm, Qpm = max(g['Ep'] for g in self.blocks), max(g['Qp'] for g in self.blocks)
        self.h, self.pooled, self.sc, self.se_h = z(M * Epm), z(B * Epm), z(B * Epm), z(B * Qpm)
        self.xh_pi, self.xh_v = z(M * self.blocks[1]['coutp']), z(M * self.blocks[2]['coutp'])
        self.hid_pi, self.logits, self.hid_v = z(B, self.Ap), z(B, self.Ap), z(B, 16)
        self.pi = torch.empty((B, self.A), dtype=f, device=d)
        self.v = torch.empty((B, self.P), dtype=f, device=d)

    def clone_buffers(self):
        import copy
        other = copy.copy(self)
        other._alloc(self.maxB)
        return other

    def _lin(self, A, lda, Wp, bias_p, out, ldc, M, K, N, act=0, R=None, ldr=0, rowscale=None, rpg=0, ksplit=False):
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        L = self._lib.lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        Kp, NP = Wp.shape
        if not ksplit and Kp // 16 in self._WS_CHUNKS:
            self._lib.check(L.azg_nn_linear_ws(p(A), lda, p(Wp), Kp, NP, p(bias_p), p(R), ldr, p(rowscale), rpg, p(out), ldc,
                                               M, K, N, act, st))
        else:
            self._lib.check(L.azg_nn_linear(p(A), lda, p(Wp), Kp, NP, p(bias_p), p(R), ldr, p(rowscale), rpg, p(out), ldc,
                                            M, K, N, act, 1 if ksplit else 0, st))

    def _block(self, g, xin, xout, B):
        import ctypes as C
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        M, Ep = B * self.L, g['Ep']
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._lin(xin, g['cinp'], g['We'], g['be'], self.h, Ep, M, g['cinp'], Ep, act=g['act'])
        self._lib.check(self._lib.lib().azg_nn_dw_pool_l(p(self.h), Ep, p(g['Wd']), p(g['sd']), p(g['bd']), p(self.pooled), B,
                                                         Ep, self.L, g['act'], g['pool_max'], st))
        self._lin(self.pooled, Ep, g['W1'], g['b1'], self.se_h, g['Qp'], B, Ep, g['Q'], act=1)
        self._lin(self.se_h, g['Qp'], g['W2'], g['b2'], self.sc, Ep, B, g['Qp'], Ep, act=3)
        self._lin(self.h, Ep, g['Wp'], g['bp'], xout, g['coutp'], M, Ep, g['cout'], R=xin if g['res'] else None,
                  ldr=g['cinp'], rowscale=self.sc, rpg=self.L)

    @torch.no_grad()
    def forward(self, boards, valids):
        import ctypes as C
        B = boards.shape[0]
        if B > self.maxB:
            self._alloc(B)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        Lb = self._lib.lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        boards = boards.reshape(B, -1)
        assert boards.dtype == torch.int8 and boards.is_contiguous() and boards.is_cuda
        valids = valids if valids.dtype == torch.uint8 else valids.to(torch.uint8)
        if self.fused:                                  # the whole forward in one launch (nn_mb1d.hip.h)
            if self.h2:
                self._lib.check(Lb.azg_nn_mb1d_forward_h2(self.geometry, p(boards), p(valids.contiguous()), self.fused_ptrs_h2,
                                                          self.descale_h2, B, p(self.pi), p(self.v), st))
            else:
                self._lib.check(Lb.azg_nn_mb1d_forward(self.geometry, p(boards), p(valids.contiguous()), self.fused_ptrs, B,
                                                       p(self.pi), p(self.v), st))
            return self.pi[:B], self.v[:B]
        self._lib.check(Lb.azg_nn_board_to_x_ld(p(boards), p(self.x0), B, self.C, self.L, self.Cp, st))
        self._lin(self.x0, self.Cp, self.pW0, self.pb0, self.x1, self.Cp, B * self.L, self.Cp, self.C)       # first_layer
        gt, gp, gv = self.blocks
        self._block(gt, self.x1, self.x2, B)
        self._block(gp, self.x2, self.xh_pi, B)
        Kp1 = self.L * gp['coutp']
        self._lin(self.xh_pi, Kp1, self.pWpi1, self.bpi1, self.hid_pi, self.Ap, B, Kp1, self.A, act=1, ksplit=True)
        self._lin(self.hid_pi, self.Ap, self.pWpi2, self.bpi2, self.logits, self.Ap, B, self.Ap, self.A, ksplit=True)
        self._block(gv, self.x2, self.xh_v, B)
        Kv1 = self.L * gv['coutp']
        self._lin(self.xh_v, Kv1, self.pWv1, self.bv1, self.hid_v, 16, B, Kv1, self.P, ksplit=True)
        self._lib.check(Lb.azg_nn_heads_out(p(self.logits), self.Ap, p(valids), p(self.hid_v), 16, p(self.Wv2), p(self.bv2),
                                            p(self.pi), p(self.v), B, self.A, self.P, st))
        return self.pi[:B], self.v[:B]

    def predict_batch(self, boards, valids):
        return self.forward(boards, valids)

    def predict(self, board, valid_actions):
        b = torch.as_tensor(np.asarray(board, dtype=np.int8)).reshape(1, -1).to(self.device)
        va = torch.as_tensor(np.asarray(valid_actions).astype(np.uint8)).reshape(1, -1).to(self.device)
        pi, v = self.forward(b, va)
        return pi[0].cpu().numpy(), v[0].cpu().numpy()


class SantoriniV89:
    """santorini/SantoriniNNet.py nn_version 88/89 (:194-219,273-281; SimpleResBlock :71-84, SimpleHead :17-40): the 2
    spatial planes (workers, levels) of the (5,5,3) board -> conv3x3(2->64)+BN+ReLU -> 5 residual blocks -> 1x1-conv heads.
    BatchNorm folded into the convolutions (eval mode); plain torch ops (MIOpen / hipBLASLt)."""

    def __init__(self, state_dict, num_players=2, device='cuda:0', dtype=torch.float32):
        sd = {k: torch.as_tensor(v).float() for k, v in state_dict.items()}
        self.P, self.A = num_players, sd['head_PI.fc.weight'].shape[0]

        def conv_bn(conv, bn):
            s, b = _fold_bn(sd, bn)
            return (sd[conv + '.weight'] * s[:, None, None, None]).contiguous(), b
        self.c0 = conv_bn('first_layer.0', 'first_layer.1')
        self.blocks = []
        i = 0
        while 'trunk.%d.conv1.weight' % i in sd:
            self.blocks.append((conv_bn('trunk.%d.conv1' % i, 'trunk.%d.bn1' % i),
                                conv_bn('trunk.%d.conv2' % i, 'trunk.%d.bn2' % i)))
            i += 1
        self.hp = conv_bn('head_PI.conv1x1', 'head_PI.bn')
        self.hv = conv_bn('head_V.conv1x1', 'head_V.bn')
        self.fc_pi = (sd['head_PI.fc.weight'].t().contiguous(), sd['head_PI.fc.bias'])
        self.fc_v1 = (sd['head_V.fc1.weight'].t().contiguous(), sd['head_V.fc1.bias'])
        self.fc_v2 = (sd['head_V.fc2.weight'].t().contiguous(), sd['head_V.fc2.bias'])
        self.to(device, dtype)

    def to(self, device, dtype=torch.float32):
        self.device, self.dtype = torch.device(device), dtype
        mv = lambda pr: tuple(t.to(self.device, dtype) for t in pr)  # noqa: E731
        self.c0, self.hp, self.hv = mv(self.c0), mv(self.hp), mv(self.hv)
        self.blocks = [(mv(a), mv(b)) for a, b in self.blocks]
        self.fc_pi, self.fc_v1, self.fc_v2 = mv(self.fc_pi), mv(self.fc_v1), mv(self.fc_v2)
        return self

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        return cls({k[3:]: z[k] for k in z.files if k.startswith('sd/')}, **kw)

    @torch.no_grad()
    def forward(self, boards, valids):
        B = boards.shape[0]
        x = boards.reshape(B, 5, 5, 3).to(self.dtype).permute(0, 3, 1, 2)[:, :2].contiguous()
        x = F.relu(F.conv2d(x, self.c0[0], self.c0[1], padding=1))
        for (w1, b1), (w2, b2) in self.blocks:
            y = F.relu(F.conv2d(x, w1, b1, padding=1))
            x = F.relu(F.conv2d(y, w2, b2, padding=1) + x)
        hp = F.relu(F.conv2d(x, self.hp[0], self.hp[1])).flatten(1)
        logits = torch.addmm(self.fc_pi[1], hp, self.fc_pi[0]).float()
        hv = F.relu(F.conv2d(x, self.hv[0], self.hv[1])).flatten(1)
        v = torch.tanh(torch.addmm(self.fc_v2[1], F.relu(torch.addmm(self.fc_v1[1], hv, self.fc_v1[0])), self.fc_v2[0]).float())
        logits = torch.where(valids.bool(), logits, torch.full_like(logits, -1e8))
        return torch.softmax(logits, dim=1).contiguous(), v.contiguous()

    def predict_batch(self, boards, valids):
        return self.forward(boards, valids)

    def predict(self, board, valid_actions):
        b = torch.from_numpy(np.ascontiguousarray(board, dtype=np.int8))[None].to(self.device)
        va = torch.from_numpy(np.asarray(valid_actions).astype(np.bool_))[None].to(self.device)
        pi, v = self.forward(b, va)
        return pi[0].cpu().numpy(), v[0].cpu().numpy()


class SantoriniV89Hip:
    """SantoriniV89 (no-gods geometry: 5 residual blocks, A = 162) evaluated by the engine's one-launch implicit-GEMM kernel
    (azg_nn_conv5_forward, csrc/nn_conv5x5.hip.h) instead of 11 MIOpen convolutions + glue ops.  Wraps a SantoriniV89."""

    def __init__(self, base, max_batch=4096, split=True, h2=True):
        """h2 (default): the trunk convolutions on f16 x 2 split-precision operands (azg_nn_conv5_forward_h2: three f16 MFMAs per
        product, 22-bit operands, same 1e-5 contract).  Otherwise split: bf16 x 3 (azg_nn_conv5_forward_split, six MFMAs per
        product); False = the f32-MFMA kernel"""
        import ctypes as C
        import math
        from . import _lib
        self._lib, self.base, self.device, self.split, self.h2 = _lib, base, base.device, bool(split), bool(h2)
        self.P, self.A = base.P, base.A
        assert base.dtype == torch.float32 and self.device.type == 'cuda' and len(base.blocks) == 5 and self.A == 162 and self.P == 2
        frag = SplendorV80Hip._frag
        d = self.device

        def conv_split(w):                 # [co][ci][3][3] -> [4 ct][18 chunks][3 planes][64 lanes][8] bf16
            co, ci = w.shape[0], w.shape[1]
            m = w.permute(2, 3, 1, 0).reshape(9 * ci, co).contiguous().float()          # [K = tap*64 + ci][co]
            hi = m.to(torch.bfloat16)
            r1 = m - hi.float()
            mid = r1.to(torch.bfloat16)
            lo = (r1 - mid.float()).to(torch.bfloat16)
            pl = torch.stack([hi, mid, lo])                                               # [3][K][co]
            pl = pl.view(3, 18, 4, 8, 4, 16)                                              # plane, chunk, g, j, ct, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                     # ct, chunk, plane, g, r, j

        def conv_mat(w, cin_pad):          # [co][ci][3][3] -> [tap*cin_pad + ci][co], fragment order
            co, ci = w.shape[0], w.shape[1]
            m = torch.zeros((9, cin_pad, co), dtype=torch.float32, device=d)
            m[:, :ci] = w.permute(2, 3, 1, 0).reshape(9, ci, co)
            return frag(m.reshape(9 * cin_pad, co).contiguous())
        convs = [c for blk in base.blocks for c in blk]
        wmax = max(float(w.abs().max()) for w, _ in convs)
        k2 = 12 - int(math.ceil(math.log2(max(wmax, 1e-30))))               # one power-of-two scale for the whole trunk: max |w| * 2^k in [2^11, 2^12]
        self.descale = (2.0 ** -k2) / 64.0

        def conv_h2(w):                    # [co][ci][3][3] -> [4 ct][18 chunks][2 planes hi, lo][64 lanes][8] f16 of W * 2^k
            co, ci = w.shape[0], w.shape[1]
            m = w.permute(2, 3, 1, 0).reshape(9 * ci, co).contiguous().float() * (2.0 ** k2)
            hi = m.to(torch.float16)
            lo = (m - hi.float()).to(torch.float16)
            pl = torch.stack([hi, lo]).view(2, 18, 4, 8, 4, 16)                          # plane, chunk, g, j, ct, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                     # ct, chunk, plane, g, r, j
        keep = [conv_mat(base.c0[0], 16), base.c0[1].contiguous(),
                torch.cat([(conv_h2(w) if self.h2 else conv_split(w) if self.split else conv_mat(w, 64)) for w, _ in convs]).contiguous(),
                torch.cat([b for _, b in convs]).contiguous(),
                base.hp[0].reshape(2, 64).t().contiguous(), base.hp[1].contiguous(), base.fc_pi[0].contiguous(), base.fc_pi[1].contiguous(),
                base.hv[0].reshape(64).contiguous(), base.hv[1].contiguous(), base.fc_v1[0].contiguous(), base.fc_v1[1].contiguous(),
                base.fc_v2[0].contiguous(), base.fc_v2[1].contiguous()]
        self._keep = keep
        self.ptrs = (C.c_void_p * 14)(*[t.data_ptr() for t in keep])
        self._alloc(max_batch)

    def _alloc(self, B):
        self.maxB = B
        self.pi = torch.empty((B, self.A), dtype=torch.float32, device=self.device)
        self.v = torch.empty((B, self.P), dtype=torch.float32, device=self.device)

    def clone_buffers(self):
        import copy
        other = copy.copy(self)
        other._alloc(self.maxB)
        return other

    @torch.no_grad()
    def forward(self, boards, valids):
        import ctypes as C
        B = boards.shape[0]
        if B > self.maxB:
            self._alloc(B)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        boards = boards.reshape(B, -1)
        assert boards.dtype == torch.int8 and boards.is_contiguous() and boards.is_cuda and boards.shape[1] == 75
        valids = (valids if valids.dtype == torch.uint8 else valids.to(torch.uint8)).contiguous()
        if self.h2:
            self._lib.check(self._lib.lib().azg_nn_conv5_forward_h2(p(boards), p(valids), self.ptrs, self.descale, 5, self.A, self.P, B, p(self.pi),
                                                                    p(self.v), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return self.pi[:B], self.v[:B]
        fwd = self._lib.lib().azg_nn_conv5_forward_split if self.split else self._lib.lib().azg_nn_conv5_forward
        self._lib.check(fwd(p(boards), p(valids), self.ptrs, 5, self.A, self.P, B, p(self.pi), p(self.v),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return self.pi[:B], self.v[:B]

    def predict_batch(self, boards, valids):
        return self.forward(boards, valids)

    def predict(self, board, valid_actions):
        b = torch.as_tensor(np.asarray(board, dtype=np.int8)).reshape(1, -1).to(self.device)
        va = torch.as_tensor(np.asarray(valid_actions).astype(np.uint8)).reshape(1, -1).to(self.device)
        pi, v = self.forward(b, va)
        return pi[0].cpu().numpy(), v[0].cpu().numpy()


class SantoriniV78(SantoriniV89):
    """santorini/SantoriniNNet.py nn_version 78 (:167-192,264-271; HeadWithMeta :42-69) -- the with-gods net of
    pretrained_withgods.pt: conv3x3(2->64, no BN) -> 10 torchvision MobileNetV3 InvertedResidual blocks (1x1 expand
    64->192 + BN + ReLU, depthwise 3x3 + BN + ReLU, 1x1 project + BN, residual; no SE) -> 1x1-conv heads whose flattened
    features are concatenated with a 32-wide embedding of the gods/metadata plane (Linear(25,32)+ReLU).  BatchNorm
    (eps 1e-5) folded into the convolutions; plain torch ops."""

    def __init__(self, state_dict, num_players=2, device='cuda:0', dtype=torch.float32):
        sd = {k: torch.as_tensor(v).float() for k, v in state_dict.items()}
        self.P, self.A = num_players, sd['head_PI.fc.weight'].shape[0]

        def conv_bn(conv, bn):
            s, b = _fold_bn(sd, bn)
            return (sd[conv + '.weight'] * s[:, None, None, None]).contiguous(), b
        self.c0 = (sd['first_layer.weight'].contiguous(), torch.zeros(sd['first_layer.weight'].shape[0]))
        self.blocks = []
        i = 0
        while 'trunk.%d.block.0.0.weight' % i in sd:
            self.blocks.append(tuple(conv_bn('trunk.%d.block.%d.0' % (i, j), 'trunk.%d.block.%d.1' % (i, j)) for j in range(3)))
            i += 1
        self.hp = conv_bn('head_PI.conv1x1', 'head_PI.bn')
        self.hv = conv_bn('head_V.conv1x1', 'head_V.bn')
        self.meta = (sd['meta_fc.1.weight'].t().contiguous(), sd['meta_fc.1.bias'])
        self.fc_pi = (sd['head_PI.fc.weight'].t().contiguous(), sd['head_PI.fc.bias'])
        self.fc_v1 = (sd['head_V.fc1.weight'].t().contiguous(), sd['head_V.fc1.bias'])
        self.fc_v2 = (sd['head_V.fc2.weight'].t().contiguous(), sd['head_V.fc2.bias'])
        self.to(device, dtype)

    def to(self, device, dtype=torch.float32):
        self.device, self.dtype = torch.device(device), dtype
        mv = lambda pr: tuple(t.to(self.device, dtype) for t in pr)  # noqa: E731
        self.c0, self.hp, self.hv, self.meta = mv(self.c0), mv(self.hp), mv(self.hv), mv(self.meta)
        self.blocks = [tuple(mv(c) for c in blk) for blk in self.blocks]
        self.fc_pi, self.fc_v1, self.fc_v2 = mv(self.fc_pi), mv(self.fc_v1), mv(self.fc_v2)
        return self

    @torch.no_grad()
    def forward(self, boards, valids):
        B = boards.shape[0]
        x = boards.reshape(B, 5, 5, 3).to(self.dtype).permute(0, 3, 1, 2)
        meta = F.relu(torch.addmm(self.meta[1], x[:, 2].reshape(B, 25), self.meta[0]))
        x = F.conv2d(x[:, :2].contiguous(), self.c0[0], self.c0[1], padding=1)
        for (we, be), (wd, bd), (wp, bp) in self.blocks:
            h = F.relu(F.conv2d(x, we, be))
            h = F.relu(F.conv2d(h, wd, bd, padding=1, groups=wd.shape[0]))
            x = F.conv2d(h, wp, bp) + x
        hp = torch.cat([F.relu(F.conv2d(x, self.hp[0], self.hp[1])).flatten(1), meta], dim=1)
        logits = torch.addmm(self.fc_pi[1], hp, self.fc_pi[0]).float()
        hv = torch.cat([F.relu(F.conv2d(x, self.hv[0], self.hv[1])).flatten(1), meta], dim=1)
        v = torch.tanh(torch.addmm(self.fc_v2[1], F.relu(torch.addmm(self.fc_v1[1], hv, self.fc_v1[0])), self.fc_v2[0]).float())
        logits = torch.where(valids.bool(), logits, torch.full_like(logits, -1e8))
        return torch.softmax(logits, dim=1).contiguous(), v.contiguous()



class SantoriniV78Hip(SantoriniV89Hip):
    """SantoriniV78 (with gods: 10 InvertedResidual blocks, A = 1782) evaluated by the engine's kernels (azg_nn_s78_forward,
    csrc/nn_conv5x5.hip.h): one launch for the trunk (MFMA GEMMs for the 1x1 convolutions, in-place depthwise 3x3 on the LDS
    tile) and the value head, one for the 132 x 1782 policy FC (MFMA, 16 samples per workgroup) + masked softmax.  Wraps a SantoriniV78."""

    def __init__(self, base, max_batch=4096, split=True, h2=True):
        """h2 (default): the 1x1 convolutions of the trunk and the depthwise pass on f16 x 2 split-precision operands
        (azg_nn_s78_forward_h2: three MFMAs per product).  Otherwise split: bf16 x 3 (azg_nn_s78_forward_split, 8 samples per
        workgroup, the expanded tile in thirds); False = the f32-MFMA kernel (4 samples per workgroup)"""
        import ctypes as C
        import math
        from . import _lib
        self._lib, self.base, self.device, self.split, self.h2 = _lib, base, base.device, bool(split) or bool(h2), bool(h2)
        self.P, self.A = base.P, base.A
        assert base.dtype == torch.float32 and self.device.type == 'cuda' and len(base.blocks) == 10 and self.A == 1782 and self.P == 2
        ke = 12 - int(math.ceil(math.log2(max(1e-30, max(float(we.abs().max()) for (we, _), _, _ in base.blocks)))))
        kp = 12 - int(math.ceil(math.log2(max(1e-30, max(float(wp.abs().max()) for _, _, (wp, _) in base.blocks)))))
        self.ds_e, self.ds_p = (2.0 ** -ke) / 64.0, (2.0 ** -kp) / 64.0

        def h2_64(m, k):                   # [64 K][64 N] f32 -> [4 ct][2 chunks of 32][2 planes hi, lo][64 lanes][8] f16 of W * 2^k
            m = m.contiguous().float() * (2.0 ** k)
            hi = m.to(torch.float16)
            lo = (m - hi.float()).to(torch.float16)
            pl = torch.stack([hi, lo]).view(2, 2, 4, 8, 4, 16)                            # plane, chunk, g, j, ct, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                     # ct, chunk, plane, g, r, j

        def split64(m):                    # [64 K][64 N] f32 -> [4 ct][2 chunks of 32][3 planes hi, mid, lo][64 lanes][8] bf16
            m = m.contiguous().float()
            hi = m.to(torch.bfloat16)
            r1 = m - hi.float()
            mid = r1.to(torch.bfloat16)
            lo = (r1 - mid.float()).to(torch.bfloat16)
            pl = torch.stack([hi, mid, lo]).view(3, 2, 4, 8, 4, 16)                       # plane, chunk, g, j, ct, r
            return pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1)                     # ct, chunk, plane, g, r, j

        def thirds(ms, of_k):              # the three 64 x 64 pieces of each [64][192] expand / [192][64] project matrix
            pack = (lambda m: h2_64(m, kp if of_k else ke)) if self.h2 else split64
            return torch.cat([pack(m[64 * t:64 * t + 64] if of_k else m[:, 64 * t:64 * t + 64]) for m in ms for t in range(3)]).contiguous()
        frag = SplendorV80Hip._frag
        d = self.device
        m0 = torch.zeros((9, 16, 64), dtype=torch.float32, device=d)
        m0[:, :2] = base.c0[0].permute(2, 3, 1, 0).reshape(9, 2, 64)
        assert float(base.c0[1].abs().max()) == 0.0          # the first conv of V78 has no bias and no BatchNorm
        cat = lambda ts: torch.cat([t.reshape(-1) for t in ts]).contiguous()  # noqa: E731
        wfp = torch.zeros((144, 1792), dtype=torch.float32, device=d)      # the policy FC for k_s78_policy: K 132 -> 144, N 1782 -> 1792
        wfp[:132, :1782] = base.fc_pi[0]
        bfp = torch.zeros(1792, dtype=torch.float32, device=d)
        bfp[:1782] = base.fc_pi[1]
        if self.h2:                        # the policy FC on f16 x 2 operands too (k_s78_policy_h2): K 132 -> 160, hi / lo fragments + the descale
            kf = 12 - int(math.ceil(math.log2(max(1e-30, float(base.fc_pi[0].abs().max())))))
            m = torch.zeros((160, 1792), dtype=torch.float32, device=d)
            m[:132, :1782] = base.fc_pi[0]
            m *= 2.0 ** kf
            hi = m.to(torch.float16)
            lo = (m - hi.float()).to(torch.float16)
            pl = torch.stack([hi, lo]).view(2, 5, 4, 8, 112, 16)                          # plane, chunk, g, j, ct, r
            fr = pl.permute(4, 1, 0, 2, 5, 3).contiguous().view(-1).view(torch.uint8)     # ct, chunk, plane, g, r, j
            tail = torch.tensor([(2.0 ** -kf) / 64.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=d).view(torch.uint8)
            wfp_h2 = torch.cat([fr, tail]).contiguous()
        keep = [frag(m0.reshape(144, 64).contiguous()),
                (thirds([we.reshape(192, 64).t() for (we, _), _, _ in base.blocks], False) if self.split else
                 cat([frag(we.reshape(192, 64).t().contiguous()) for (we, _), _, _ in base.blocks])), cat([be for (_, be), _, _ in base.blocks]),
                cat([wd.reshape(192, 9) for _, (wd, _), _ in base.blocks]), cat([bd for _, (_, bd), _ in base.blocks]),
                (thirds([wp.reshape(64, 192).t() for _, _, (wp, _) in base.blocks], True) if self.split else
                 cat([frag(wp.reshape(64, 192).t().contiguous()) for _, _, (wp, _) in base.blocks])), cat([bp for _, _, (_, bp) in base.blocks]),
                base.meta[0].contiguous(), base.meta[1].contiguous(),
                base.hp[0].reshape(4, 64).t().contiguous(), base.hp[1].contiguous(), (wfp_h2 if self.h2 else frag(wfp)), bfp,
                base.hv[0].reshape(2, 64).t().contiguous(), base.hv[1].contiguous(), base.fc_v1[0].contiguous(), base.fc_v1[1].contiguous(),
                base.fc_v2[0].contiguous(), base.fc_v2[1].contiguous()]
        assert len(keep) == 19 and tuple(base.fc_pi[0].shape) == (132, 1782) and tuple(base.fc_v1[0].shape) == (82, 64)
        self._keep = keep
        self.ptrs = (C.c_void_p * 19)(*[t.data_ptr() for t in keep])
        self._alloc(max_batch)

    @torch.no_grad()
    def forward(self, boards, valids):
        import ctypes as C
        B = boards.shape[0]
        if B > self.maxB:
            self._alloc(B)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        boards = boards.reshape(B, -1)
        assert boards.dtype == torch.int8 and boards.is_contiguous() and boards.is_cuda and boards.shape[1] == 75
        valids = (valids if valids.dtype == torch.uint8 else valids.to(torch.uint8)).contiguous()
        if self.h2:
            self._lib.check(self._lib.lib().azg_nn_s78_forward_h2(p(boards), p(valids), self.ptrs, self.ds_e, self.ds_p, 10, self.A, self.P, B,
                                                                  p(self.pi), p(self.v), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return self.pi[:B], self.v[:B]
        fwd = self._lib.lib().azg_nn_s78_forward_split if self.split else self._lib.lib().azg_nn_s78_forward
        self._lib.check(fwd(p(boards), p(valids), self.ptrs, 10, self.A, self.P, B, p(self.pi), p(self.v),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return self.pi[:B], self.v[:B]



class TorchModuleEvaluator:
    """Leaf evaluator around ANY torch module with the reference's forward signature
    `module(board f32[B, *board_shape], valid_actions bool[B, A]) -> (log_pi f32[B, A], v f32[B, P])` -- the torch branch of
    GenericNNetWrapper.predict (:111-120) batched on PyTorch-ROCm.  It is the evaluator of the games whose nets have no engine
    kernel: the six f4 games with the reference's own `<G>NNet` modules (import them from the reference, load their checkpoints
    as usual) or a user's architecture.  Same `predict_batch` contract as the engine-kernel nets, so SelfPlayEngine / BatchedMCTS /
    BatchedArena / Coach take it unchanged (HIP-graph capture included: the module runs inside the captured round)."""

    def __init__(self, module, game, max_batch=None):
        self.module = module.to(game.device).eval()
        self.shape = tuple(game.getBoardSize())
        self.device = game.device

    def predict_batch(self, boards, valids):
        with torch.no_grad():
            b = boards.reshape((boards.shape[0],) + self.shape).to(torch.float32)
            log_pi, v = self.module(b, valids.bool())
            return torch.exp(log_pi).to(torch.float32).contiguous(), v.to(torch.float32).contiguous()

    def clone_buffers(self):
        return self
