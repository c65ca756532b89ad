"""Integer hash-net of SURVEY.md Appendix C.3 as a batched torch module (bit-reproducible on any backend): the
deterministic NeuralNet.predict used on both sides of the MCTS parity tests."""
import torch


class HashNetTorch:
    def __init__(self, P):
        self.P = P

    def predict_batch(self, boards, valids):
        T = boards.shape[0]
        flat = boards.reshape(T, -1).to(torch.int64)
        idx = torch.arange(1, flat.shape[1] + 1, dtype=torch.int64, device=flat.device)
        s = (flat * idx).sum(dim=1)
        h = torch.remainder(s * 2654435761, 1 << 32)
        v0 = (h.to(torch.float64) / 2147483648.0 - 1.0).to(torch.float32)
        others = (-v0.to(torch.float64) / float(self.P - 1)).to(torch.float32)   # f64 divide: torch's f32 scalar divide multiplies by a reciprocal
        v = torch.stack([v0] + [others] * (self.P - 1), dim=1)
        A = valids.shape[1]
        a = torch.arange(A, dtype=torch.int64, device=flat.device)
        w = valids.to(torch.int64) * (1 + torch.remainder(torch.remainder((h >> 8)[:, None] + 2654435761 * a[None, :],
                                                                          1 << 32), 13))
        pi = (w.to(torch.float64) / w.sum(dim=1, keepdim=True).to(torch.float64)).to(torch.float32)
        return pi.contiguous(), v.contiguous()


class HashNetNumpy:
    """the same integer hash-net behind the reference's per-sample NeuralNet.predict(board, valid_actions) signature
    (NeuralNet.py:32-43): numpy in, (pi f32[A], v f32[P]) out -- exercises the engine's single-sample slow path"""

    def __init__(self, P):
        self.P, self.calls = P, 0

    def predict(self, board, valids):
        import numpy as np
        self.calls += 1
        flat = np.asarray(board).reshape(-1).astype(np.int64)
        s = int((flat * np.arange(1, flat.size + 1, dtype=np.int64)).sum())
        h = (s * 2654435761) % (1 << 32)
        v0 = np.float32(h / 2147483648.0 - 1.0)
        v = np.array([v0] + [np.float32(-float(v0) / (self.P - 1))] * (self.P - 1), dtype=np.float32)
        a = np.arange(len(valids), dtype=np.int64)
        w = np.asarray(valids).astype(np.int64) * (1 + (((h >> 8) + 2654435761 * a) % (1 << 32)) % 13)
        return (w / w.sum()).astype(np.float32), v


class HashNetHip:
    """the same hash-net evaluated by the engine library (azg_eval_hashnet, one wave per sample): bit-identical to HashNetTorch, one
    launch instead of ~35 torch kernels -- the evaluator of tools/bench_f4.py --net hashhip"""

    def __init__(self, P):
        self.P = P
        self._out = {}

    def bind_outputs(self, T, A, P, device):
        """the engine announces the batch it will evaluate: fixed output buffers `pi` / `v`, which selfplay._Group then hands to the
        expansion as they are (no copy per round: two 7-us copy kernels per 50-us round in tools/bench_f4.py --net hashhip)"""
        self.pi = torch.empty((T, A), dtype=torch.float32, device=device)
        self.v = torch.empty((T, P), dtype=torch.float32, device=device)
        self._out[(T, A, torch.device(device))] = (self.pi, self.v)

    def predict_batch(self, boards, valids):
        import ctypes as C
        from azg_amd import _lib
        T, A = valids.shape
        S = boards[0].numel()
        key = (T, A, boards.device)
        if key not in self._out:
            self._out[key] = (torch.empty((T, A), dtype=torch.float32, device=boards.device), torch.empty((T, self.P), dtype=torch.float32, device=boards.device))
        pi, v = self._out[key]
        b = boards if boards.is_contiguous() else boards.contiguous()
        va = valids if valids.dtype == torch.uint8 and valids.is_contiguous() else valids.to(torch.uint8).contiguous()
        _lib.check(_lib.lib().azg_eval_hashnet(C.c_void_p(b.data_ptr()), C.c_void_p(va.data_ptr()), T, S, A, self.P, C.c_void_p(pi.data_ptr()),
                                               C.c_void_p(v.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return pi, v


class HashNetPipeline(HashNetHip):
    """the hash-net as the evaluator of the asynchronous tree pipeline (azg_forest_async_rounds_hashnet: evaluated INSIDE the pipeline's
    persistent evaluator kernel); as a batched evaluator it is HashNetHip, so the same object also drives the two-kernel rounds"""
    async_hashnet = True
