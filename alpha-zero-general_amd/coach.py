"""Next-row f1: Coach (Coach.py:20-263) on the engine -- same constructor and method names:

    c = Coach(game, nnet, args)         # nnet: azg_amd.nnet_wrapper.NNetWrapper (or a bare trainable module, wrapped here)
    c.loadTrainExamples(); c.learn(); c.executeEpisodes(); c.saveTrainExamples(); c.getCheckpointFile(i); c.temp_for_game(n)

Self-play runs on the GPU forest (SelfPlayEngine: exactly numEps episodes, each played to its end, fresh randomness every
iteration), training on PyTorch-ROCm autograd (train.py), the accept / reject gate on BatchedArena, files in the reference's
formats and names (checkpoint.examples, temp.pt, checkpoint_<i>.pt, best.pt).  Host-side orchestration only; every hot loop is
one of the pieces above.  Engine-only knobs (concurrent games, node capacity) are keyword arguments or `args.n_games`."""
import os
import pickle
import random
import sys
import zlib

from . import formats
from .arena import BatchedArena
from .nnet_wrapper import NNetWrapper
from .selfplay import SelfPlayEngine


def _get(args, k, d):
    v = args.get(k, d) if isinstance(args, dict) else getattr(args, k, d)
    return d if v is None else v


def _vars(args):
    return dict(args) if isinstance(args, dict) else dict(vars(args))


class Coach:
    def __init__(self, game, nnet, args, n_games=None, node_capacity=None, log=print):
        self.game, self.args, self.log = game, args, log
        if not hasattr(nnet, 'save_checkpoint'):                     # a bare torch module: give it the NeuralNet surface
            w = NNetWrapper(game, dict(nn_version=getattr(nnet, 'version', -1), learn_rate=_get(args, 'learn_rate', 3e-3),
                                       batch_size=_get(args, 'batch_size', 512), epochs=_get(args, 'epochs', 2),
                                       q_weight=_get(args, 'q_weight', 0.5), dropout=_get(args, 'dropout', 0.0)))
            w.nnet = nnet
            nnet = w
        self.nnet = nnet
        self.pnet = self.nnet.__class__(self.game, self.nnet.args)                      # the competitor network (Coach.py:30)
        self.T = int(n_games or _get(args, 'n_games', 0) or max(1, min(4096, int(_get(args, 'numEps', 256)))))
        self.cap = node_capacity
        self.trainExamplesHistory = []                                                  # Coach.py:32
        self.skipFirstSelfPlay = bool(getattr(nnet, 'requestKnowledgeTransfer', False))
        self.consecutive_failures = 0
        self.engine = None
        self.n_selfplay_waves = 0
        self.results = []

    # engine-side aliases kept from round 1
    @property
    def module(self):
        return self.nnet.nnet

    @property
    def history(self):
        return self.trainExamplesHistory

    def execute_episodes(self):
        return self.executeEpisodes(as_tensors=True)

    # ---- Coach.executeEpisodes (:86-148) ----
    def executeEpisodes(self, as_tensors=False):
        """numEps finished games of self-play with the current net -> one iteration's examples (a deque in the reference's
        layout; as_tensors=True: the device tensors of SelfPlayEngine.drain_examples(symmetries=True))"""
        num_eps = int(_get(self.args, 'numEps', self.T))
        net = self.nnet.evaluator(self.T)
        if self.engine is None:
            self.engine = SelfPlayEngine(self.game, net, self.args, self.T, node_capacity=self.cap,
                                         max_examples=max(self.T, num_eps) * 160, rng_seed=int(_get(self.args, 'seed', 0)))
        else:
            for grp in self.engine.groups:
                grp.net = net
            self.engine.nnet, self.engine.graph = net, None          # new weights: the captured rounds are stale
        # exactly numEps episodes, each played to its end (Coach.py:86-148); a new RNG epoch per call, so an iteration never
        # replays the games of the previous one (the reference draws fresh randomness every iteration)
        self.n_selfplay_waves += 1
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
        ex = self.engine.drain_examples(symmetries=True)
        if as_tensors:
            return ex
        return formats.examples_to_iteration(ex, tuple(self.game.getBoardSize()),
                                             compress=not _get(self.args, 'no_compression', False),
                                             maxlen=int(_get(self.args, 'maxlenOfQueue', 10 ** 6)))

    # ---- Coach.learn (:150-215) ----
    def learn(self):
        a = self.args
        ckpt = _get(a, 'checkpoint', './checkpoint')
        os.makedirs(ckpt, exist_ok=True)
        for i in range(1, int(_get(a, 'numIters', 1)) + 1):
            if not self.skipFirstSelfPlay or i > 1:
                it_examples = self.executeEpisodes()
                if len(it_examples) == int(_get(a, 'maxlenOfQueue', 10 ** 6)):
                    self.log('saturation of elements in iterationTrainExamples, think about decreasing numEps or increasing maxlenOfQueue')
                self.trainExamplesHistory.append(it_examples)
            if _get(a, 'profile', False):
                return self.results
            if len(self.trainExamplesHistory) > int(_get(a, 'numItersHistory', 5)):
                self.trainExamplesHistory.pop(0)
            self.saveTrainExamples()                                                                  # :180
            train_examples = [e for it in self.trainExamplesHistory for e in it]
            random.shuffle(train_examples)                                                            # :185
            extra = {k: v for k, v in _vars(a).items() if isinstance(v, (int, float, bool, str, list, tuple))}
            self.nnet.save_checkpoint(folder=ckpt, filename='temp.pt', additional_keys=extra)         # :188
            self.pnet.load_checkpoint(folder=ckpt, filename='temp.pt')
            self.nnet.train(train_examples, log=self.log)
            n_arena = int(_get(a, 'arenaCompare', 30))
            # a fresh block of RNG streams per iteration (boards and chance outcomes of the arena games)
            arena = BatchedArena(self.game, self.nnet.evaluator(n_arena), self.pnet.evaluator(n_arena), a,
                                 n_parallel=n_arena, node_capacity=self.cap, stream0=(1 << 32) * i,
                                 temp_for_game=self.temp_for_game)
            nwins, pwins, draws = arena.playGames(n_arena)
            for m in arena.mcts:
                m.forest.close()
            accepted = (pwins + nwins) > 0 and float(nwins) / (pwins + nwins) >= float(_get(a, 'updateThreshold', 0.6))
            self.results.append(dict(iteration=i, examples=len(train_examples), nwins=nwins, pwins=pwins, draws=draws, accepted=accepted))
            if not accepted:
                self.consecutive_failures += 1
                self.log('Iter #%d - new vs previous: %d-%d  (%d draws) --> REJECTED (%d)' % (i, nwins, pwins, draws, self.consecutive_failures))
                if self.consecutive_failures >= int(_get(a, 'stop_after_N_fail', 10 ** 9)) and i < int(_get(a, 'numIters', 1)):
                    self.log('Exceeded threshold number of consecutive fails, stopping process')                 # :204-206
                    if _get(a, 'exit_on_fail', True):
                        sys.exit()
                    return self.results
                self.nnet.load_checkpoint(folder=ckpt, filename='temp.pt')                                        # :207
            else:
                self.log('Iter #%d - new vs previous: %d-%d  (%d draws) --> ACCEPTED' % (i, nwins, pwins, draws))
                self.nnet.save_checkpoint(folder=ckpt, filename=self.getCheckpointFile(i), additional_keys=extra)
                self.nnet.save_checkpoint(folder=ckpt, filename='best.pt', additional_keys=extra)
                self.consecutive_failures = 0
        return self.results

    def getCheckpointFile(self, iteration):
        return 'checkpoint_' + str(iteration) + '.pt'

    def saveTrainExamples(self):                                                                      # :220-226
        folder = _get(self.args, 'checkpoint', './checkpoint')
        os.makedirs(folder, exist_ok=True)
        formats.save_train_examples(os.path.join(folder, 'checkpoint.examples'), self.trainExamplesHistory)

    def loadTrainExamples(self):                                                                      # :228-263
        model_file = _get(self.args, 'load_folder_file', None)
        model_file = os.path.join(*model_file) if isinstance(model_file, (list, tuple)) else model_file
        path = os.path.join(os.path.dirname(model_file or ''), 'checkpoint.examples')
        if not os.path.isfile(path):
            self.log('File "%s" with trainExamples not found!' % path)
            return
        with open(path, 'rb') as f:
            self.trainExamplesHistory = pickle.load(f)
        # harmonise the compression with args.no_compression (:243-250)
        want_raw = bool(_get(self.args, 'no_compression', False))
        for it in self.trainExamplesHistory:
            for j in range(len(it)):
                if isinstance(it[j], tuple) and not want_raw:
                    it[j] = zlib.compress(pickle.dumps(it[j]), level=1)
                elif not isinstance(it[j], tuple) and want_raw:
                    it[j] = pickle.loads(zlib.decompress(it[j]))
        n_hist, maxlen = int(_get(self.args, 'numItersHistory', 5)), int(_get(self.args, 'maxlenOfQueue', 10 ** 6))
        if len(self.trainExamplesHistory) > n_hist:                                                   # :254-256
            self.trainExamplesHistory = self.trainExamplesHistory[-n_hist:]
        for it in self.trainExamplesHistory:                                                          # :257-261
            while len(it) > maxlen:
                it.pop()

    def temp_for_game(self, n):                                                                       # :273-276
        t_begin, t_end, half_life = 0.5, 0.0, abs(float(_get(self.args, 'tempThreshold', 10)))
        return t_end + (t_begin - t_end) * (0.5 ** (n / half_life))

    # round-1 name: GenericNNetWrapper.save_checkpoint of an arbitrary module with this Coach's args embedded
    def save_checkpoint(self, folder, filename, module):
        w = NNetWrapper.__new__(NNetWrapper)
        w.nnet = module
        extra = {k: v for k, v in _vars(self.args).items() if isinstance(v, (int, float, bool, str, list, tuple))}
        NNetWrapper.save_checkpoint(w, folder, filename, extra)
