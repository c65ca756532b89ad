"""Coach.executeEpisodes on the engine (Coach.py:86-148): T games per GPU, device-resident episode state machines, one
batched NeuralNet.predict per lock-step round.  The reference time-slices N game threads on one core around a lock ring
to build inference batches of N (Coach.py:117-144); here every round is
    select (HIP) -> predict_batch (PyTorch-ROCm, MFMA GEMMs) -> expand_backup (HIP) -> selfplay_advance (HIP)
with no host decision in the loop, so it can be captured once in a HIP graph and replayed."""
import torch

from .forest import Forest


class SelfPlayEngine:
    def __init__(self, game, nnet, args, n_games, node_capacity=None, max_examples=None, rng_seed=0, stream0=0,
                 use_graph=True, dirichlet=None):
        self.game, self.nnet, self.args = game, nnet, args
        get = (lambda k, d: args.get(k, d)) if isinstance(args, dict) else (lambda k, d: getattr(args, k, d))
        sims = int(get('numMCTSSims', 800))
        cap = node_capacity or max(1024, 8 * sims)
        self.T = n_games
        self.forest = Forest(game.GAME_ID, game.variant, n_games, args, node_capacity=cap,
                             max_examples=max_examples or n_games * 64, rng_seed=rng_seed, stream0=stream0,
                             device=str(game.device))
        self.alpha = float(get('dirichletAlpha', 0.0)) if dirichlet is None else float(dirichlet)
        self.shape = (n_games,) + tuple(self.forest.board_shape())
        dev = self.forest.device
        # Coach passes dirichlet_noise=(dirichletAlpha != 0) (Coach.py:31,96).  The Gamma variates of
        # rng.dirichlet([alpha]*n_valid) (MCTS.py:187-192) are drawn on device by the engine's own sampler.
        self.device_noise = self.alpha != 0.0
        self.use_graph = use_graph
        self.graph = None
        self.rounds = 0
        self._pi = self._v = None

    def start(self, init_boards=None):
        self.forest.selfplay_start(init_boards)
        torch.cuda.synchronize()

    def _round(self):
        f = self.forest
        f.select(device_noise=self.device_noise)
        pi, v = self.nnet.predict_batch(f.leaf_states.view(self.shape), f.leaf_valid)
        self._pi, self._v = pi, v
        f.expand_backup(pi, v, device_noise=self.device_noise)
        f.selfplay_advance()

    def capture(self):
        """Capture one round in a HIP graph (after a few eager warm-up rounds on a side stream)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self._round()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._round()
        self.graph = g
        self.rounds += 4

    def run(self, rounds):
        if self.use_graph and self.graph is None:
            self.capture()
        if self.graph is not None:
            for _ in range(rounds):
                self.graph.replay()
        else:
            for _ in range(rounds):
                self._round()
        self.rounds += rounds

    def stats(self):
        return self.forest.stats()

    def drain_examples(self):
        """-> (boards int8[n,S], pi f32[n,A], z f32[n,P], valids u8[n,A], q f32[n,P], meta i32[n,4]) of finished games
        (Coach.py:76-82 record layout, un-augmented; symmetries are applied by the consumer)."""
        return self.forest.drain_examples()


def gather_examples(tensors, group=None):
    """Multi-GPU episode-end gather of variable-length example records over RCCL (torch.distributed 'nccl' backend on
    ROCm): all_gather the counts, pad to the maximum, all_gather the padded blocks, trim.  With games sharded
    embarrassingly this is the ONLY collective on the path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    world = dist.get_world_size(group)
    n = torch.tensor([tensors[0].shape[0]], dtype=torch.int64, device=tensors[0].device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    out = []
    for t in tensors:
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out.append(torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0))
    return out
