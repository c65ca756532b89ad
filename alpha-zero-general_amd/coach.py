"""Next-row f1: Coach.learn (Coach.py:150-215) on the engine -- self-play on the GPU forest
(SelfPlayEngine), training with PyTorch-ROCm autograd (train.py), the accept / reject gate played by BatchedArena, the
reference's examples file and checkpoint names.  Host-side orchestration only; every hot loop is one of the pieces above."""
import copy
import os

import torch

from . import formats
from .arena import BatchedArena
from .nnet import SplendorV80Hip
from .selfplay import SelfPlayEngine
from .train import train


def _get(args, k, d):
    v = args.get(k, d) if isinstance(args, dict) else getattr(args, k, d)
    return d if v is None else v


class Coach:
    def __init__(self, game, module, args, n_games=256, node_capacity=None, log=print):
        self.game, self.module, self.args, self.T, self.cap, self.log = game, module, args, n_games, node_capacity, log
        self.history = []                                            # trainExamplesHistory (Coach.py:32)
        self.consecutive_failures = 0
        self.engine = None
        self.results = []

    def _infer_net(self, module, max_batch):
        """the engine-kernel evaluator of the module's current weights (BatchNorm folded): the 2-player V80 layout kernel, or
        the generic one-launch kernel for the other geometries of the family (Splendor 3-4p, Azul V84)"""
        from . import nnet
        sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
        dev = str(self.game.device)
        if getattr(module, 'version', 80) == 84:
            return nnet.MobileNet1dHip(nnet.AzulV84(sd, num_players=self.game.P, device=dev), max_batch=max_batch)
        if self.game.P == 2:
            return SplendorV80Hip(sd, num_players=2, device=dev, max_batch=max_batch)
        return nnet.MobileNet1dHip(nnet.SplendorV80(sd, num_players=self.game.P, device=dev), max_batch=max_batch)

    def execute_episodes(self):
        """Coach.executeEpisodes (:86-148): numEps finished games of self-play with the current net -> one iteration's examples"""
        num_eps = int(_get(self.args, 'numEps', self.T))
        net = self._infer_net(self.module, self.T)
        if self.engine is None:
            self.engine = SelfPlayEngine(self.game, net, self.args, self.T, node_capacity=self.cap,
                                         max_examples=max(self.T, num_eps) * 160, rng_seed=int(_get(self.args, 'seed', 0)))
        else:
            for grp in self.engine.groups:
                grp.net = net
            self.engine.nnet, self.engine.graph = net, None          # new weights: the captured rounds are stale
        # exactly numEps episodes, each played to its end (Coach.py:86-148); a new RNG epoch per call, so an iteration never
        # replays the games of the previous one (the reference draws fresh randomness every iteration)
        self.n_selfplay_waves = getattr(self, 'n_selfplay_waves', 0) + 1
        self.engine.start(epoch=self.n_selfplay_waves, episode_quota=num_eps)
        sims = int(_get(self.args, 'numMCTSSims', 800))
        while True:
            self.engine.run(8 * max(8, sims))
            st = self.engine.stats()
            if st['errors']:
                raise RuntimeError('engine error flags %d (16 = example ring overflow: %d records dropped)'
                                   % (st['errors'], st['examples_dropped']))
            if st['active'] == 0:
                break
        assert st['games'] == num_eps, (st['games'], num_eps)
        return self.engine.drain_examples(symmetries=True)

    def learn(self):
        a = self.args
        ckpt = _get(a, 'checkpoint', './checkpoint')
        os.makedirs(ckpt, exist_ok=True)
        for it in range(1, int(_get(a, 'numIters', 1)) + 1):
            ex = self.execute_episodes()
            self.history.append(formats.examples_to_iteration(ex, tuple(self.game.getBoardSize()),
                                                              compress=not _get(a, 'no_compression', False),
                                                              maxlen=int(_get(a, 'maxlenOfQueue', 10 ** 6))))
            if len(self.history) > int(_get(a, 'numItersHistory', 5)):
                self.history.pop(0)
            formats.save_train_examples(os.path.join(ckpt, 'checkpoint.examples'), self.history)      # Coach.py:180
            flat = [e for itx in formats.load_train_examples(os.path.join(ckpt, 'checkpoint.examples')) for e in itx]
            import numpy as np
            cols = [np.stack([np.asarray(e[k]).reshape(-1) for e in flat]) for k in range(5)]
            self.save_checkpoint(ckpt, 'temp.pt', self.module)
            previous = copy.deepcopy(self.module)
            train(self.module, cols, learn_rate=float(_get(a, 'learn_rate', 3e-3)), batch_size=int(_get(a, 'batch_size', 512)),
                  epochs=int(_get(a, 'epochs', 2)), q_weight=float(_get(a, 'q_weight', 0.5)), device=str(self.game.device),
                  log=self.log)
            n_arena = int(_get(a, 'arenaCompare', 30))
            # a fresh block of RNG streams per iteration (boards and chance outcomes of the arena games)
            arena = BatchedArena(self.game, self._infer_net(self.module, n_arena), self._infer_net(previous, n_arena), a,
                                 n_parallel=n_arena, node_capacity=self.cap, stream0=(1 << 32) * it)
            nwins, pwins, draws = arena.playGames(n_arena)
            for m in arena.mcts:
                m.forest.close()
            accepted = (pwins + nwins) > 0 and float(nwins) / (pwins + nwins) >= float(_get(a, 'updateThreshold', 0.6))
            self.results.append(dict(iteration=it, examples=len(flat), nwins=nwins, pwins=pwins, draws=draws, accepted=accepted))
            self.log('Iter #%d - new vs previous: %d-%d (%d draws) --> %s' % (it, nwins, pwins, draws,
                                                                              'ACCEPTED' if accepted else 'REJECTED'))
            if accepted:
                self.consecutive_failures = 0
                self.save_checkpoint(ckpt, 'checkpoint_%d.pt' % it, self.module)
                self.save_checkpoint(ckpt, 'best.pt', self.module)
            else:
                self.consecutive_failures += 1
                self.module.load_state_dict(previous.state_dict())                                   # Coach.py:202
        return self.results

    def save_checkpoint(self, folder, filename, module):
        """GenericNNetWrapper.save_checkpoint (:192-205): state_dict + full_model + the args as extra keys"""
        data = {'state_dict': {k: v.detach().cpu() for k, v in module.state_dict().items()}, 'full_model': copy.deepcopy(module).cpu()}
        extra = dict(self.args) if isinstance(self.args, dict) else dict(vars(self.args))
        data.update({k: v for k, v in extra.items() if isinstance(v, (int, float, bool, str, list, tuple))})
        torch.save(data, os.path.join(folder, filename))
