"""MCTS.py-compatible search on the HIP forest.

`MCTS(game, nnet, args, dirichlet_noise=False)` keeps the reference's constructor and `getActionProb` signature
(MCTS.py:24,49) for ONE tree; `BatchedMCTS` runs T trees in lock-step: every round each tree contributes one leaf to
a single NeuralNet.predict-shaped batch (what the reference's thread ring does with --parallel-inferences,
Coach.py:117-144, GenericNNetWrapper.py:122-157).

nnet must offer either `predict_batch(boards int8[T,...] cuda, valids bool[T,A] cuda) -> (pi f32[T,A], v f32[T,P])`
(probabilities, not log-probabilities: GenericNNetWrapper.py:107,119) or the reference's per-sample `predict`."""
import numpy as np
import torch

from .forest import Forest


class _ArgsView:
    """the caller's args with a few values overridden (the original object is not touched)"""

    def __init__(self, base, **over):
        self._b, self._o = base, over

    def __getattr__(self, k):
        if k.startswith('_'):
            raise AttributeError(k)
        if k in self._o:
            return self._o[k]
        b = self._b
        if isinstance(b, dict):
            if k in b:
                return b[k]
            raise AttributeError(k)
        return getattr(b, k)

    def get(self, k, d=None):
        if k in self._o:
            return self._o[k]
        b = self._b
        v = b.get(k, d) if isinstance(b, dict) else getattr(b, k, d)
        return d if v is None else v


class BatchedMCTS:
    def __init__(self, game, nnet, args, n_trees, dirichlet_noise=False, node_capacity=None, **forest_kw):
        self.game, self.nnet, self.args = game, nnet, args
        self.dirichlet_noise = bool(dirichlet_noise)
        if not self.dirichlet_noise:
            # no root noise for this MCTS object whatever args.dirichletAlpha says (MCTS.py:64): the forest applies noise
            # only when its own alpha is non-zero
            args = _ArgsView(args, dirichletAlpha=0.0)
        sims = int(getattr(args, 'numMCTSSims', 800) if not isinstance(args, dict) else args.get('numMCTSSims', 800))
        # nodes live until the root's round passes theirs (the clean-up before each search, cf. MCTS.py:86-91): a few
        # plies' worth of simulations
        cap = node_capacity or max(1024, 10 * sims + 512)
        self.forest = Forest(game.GAME_ID, game.variant, n_trees, args, node_capacity=cap, device=str(game.device),
                             **forest_kw)
        self.T = n_trees

    def _predict(self, boards, valids):
        if hasattr(self.nnet, 'predict_batch'):
            pi, v = self.nnet.predict_batch(boards.view((self.T,) + tuple(self.game.getBoardSize())), valids.bool())
            return pi.float().contiguous(), v.float().contiguous()
        pis, vs = [], []
        b, va = boards.cpu().numpy(), valids.cpu().numpy().astype(bool)
        for i in range(self.T):
            pi, v = self.nnet.predict(b[i].reshape(self.game.getBoardSize()), va[i])
            pis.append(pi); vs.append(v)
        dev = boards.device
        return (torch.tensor(np.asarray(pis), dtype=torch.float32, device=dev),
                torch.tensor(np.asarray(vs), dtype=torch.float32, device=dev))

    def search(self, roots, full=None, noise=None):
        """Run numMCTSSims simulations from `roots` (int8 cuda tensor [T, S...])."""
        f = self.forest
        f.begin_search(roots.reshape(self.T, -1), full)
        # root Dirichlet noise (MCTS.py:64,147-149,156-160,187-197): on simulation 0 of a FULL search of an MCTS built with
        # dirichlet_noise=True (Coach.py:31,96 passes dirichletAlpha != 0; Arena / pit players pass nothing).  `noise`
        # injects the sample (parity tests); otherwise the engine draws rng.dirichlet([alpha] * n_valid) on device.
        dev_noise = bool(self.dirichlet_noise) and noise is None and float(f.cfg.dirichletAlpha) != 0.0
        if noise is not None:
            assert self.dirichlet_noise, 'a noise sample was passed to an MCTS built with dirichlet_noise=False'
        rounds = 0
        while True:
            f.select(noise, device_noise=dev_noise)
            ne = f.needs_eval
            if not bool(ne.any().item()):
                if f.active() == 0:
                    break
                continue
            pi, v = self._predict(f.leaf_states, f.leaf_valid)
            f.expand_backup(pi, v, noise, device_noise=dev_noise)
            rounds += 1
        return rounds

    def getActionProb(self, canonicalBoards, temp=1, force_full_search=False, full=None, noise=None):
        dev = self.forest.device
        roots = canonicalBoards if torch.is_tensor(canonicalBoards) else torch.from_numpy(
            np.ascontiguousarray(canonicalBoards, dtype=np.int8))
        roots = roots.to(dev).reshape(self.T, -1)
        if full is None:
            p_full = float(self.forest.cfg.prob_fullMCTS)
            if force_full_search or p_full >= 1.0:
                full = torch.ones(self.T, dtype=torch.uint8, device=dev)
            else:
                full = (torch.rand(self.T, device=dev) < p_full).to(torch.uint8)       # MCTS.py:58
        self.search(roots, full, noise)
        probs, q, is_full = self.forest.action_probs(temp)
        return probs, q, is_full

    def reset_all_search_trees(self):
        self.forest.reset()


class MCTS:
    """Drop-in for the reference's MCTS class (one tree)."""

    def __init__(self, game, nnet, args, dirichlet_noise=False, batch_info=None):
        self.game, self.nnet, self.args = game, nnet, args
        self._b = BatchedMCTS(game, nnet, args, 1, dirichlet_noise=dirichlet_noise)
        self.step = 0

    def getActionProb(self, canonicalBoard, temp=1, force_full_search=False):
        probs, q, full = self._b.getActionProb(np.asarray(canonicalBoard, dtype=np.int8)[None], temp, force_full_search)
        return list(probs[0].cpu().numpy()), list(q[0].cpu().numpy()), bool(full[0].item())

    @staticmethod
    def reset_all_search_trees():
        import gc
        for o in [o for o in gc.get_objects() if type(o) is MCTS]:
            o._b.reset_all_search_trees()
