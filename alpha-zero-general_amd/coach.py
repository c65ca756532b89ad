"""Next-row f1: Coach (Coach.py:20-263) on the engine -- same constructor and method names:

    c = Coach(game, nnet, args)         # nnet: azg_amd.nnet_wrapper.NNetWrapper (or a bare trainable module, wrapped here)
    c.loadTrainExamples(); c.learn(); c.executeEpisodes(); c.saveTrainExamples(); c.getCheckpointFile(i); c.temp_for_game(n)

Self-play runs on the GPU forest (SelfPlayEngine: exactly numEps episodes, each played to its end, fresh randomness every
iteration), training on PyTorch-ROCm autograd (train.py), the accept / reject gate on BatchedArena, files in the reference's
formats and names (checkpoint.examples, temp.pt, checkpoint_<i>.pt, best.pt).  Host-side orchestration only; every hot loop is
one of the pieces above.  Engine-only knobs (concurrent games, node capacity) are keyword arguments or `args.n_games`.

Several GPUs (BASELINE config 5): with torch.distributed initialised (one process per GPU) every rank owns a
contiguous range of the n_games concurrent game streams and of the numEps episodes, the finished games' records are exchanged
once per iteration (selfplay.gather_examples: all_gather of the counts + ONE grouped send / receive of the packed records to rank 0), rank 0 trains and writes the files, the new
weights are broadcast, the arena games are sharded by game index and the three tallies all_reduced -- every rank takes the same
accept / reject decision.  The global game streams, the episodes each stream plays and the arena games do not depend on the world
size, and the gathered records are put in a canonical order before training, so a run on W ranks reproduces the run on one."""
import copy
import json
import os
import pickle
import random
import sys
import zlib

import torch

from . import formats
from .arena import BatchedArena
from .nnet_wrapper import NNetWrapper
from .selfplay import SelfPlayEngine


def _get(args, k, d):
    v = args.get(k, d) if isinstance(args, dict) else getattr(args, k, d)
    return d if v is None else v


def _vars(args):
    return dict(args) if isinstance(args, dict) else dict(vars(args))


def split_range(n, world, rank):
    """contiguous share of n items for `rank` of `world`: (first, count); the first n % world ranks get one more"""
    base, rem = divmod(int(n), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def episode_share(num_eps, n_games, world, rank):
    """Episodes of rank `rank` when the n_games global game streams are dealt out in contiguous blocks of n_games / world: global
    stream t plays num_eps // n_games (+1 for t < num_eps % n_games) games (include/azg.h azg_selfplay_start_ex), a rank plays
    what its streams play -- the same games whatever the world size"""
    tl = n_games // world
    a, b = divmod(int(num_eps), int(n_games))
    return a * tl + max(0, min(tl, b - rank * tl))


class Coach:
    def __init__(self, game, nnet, args, n_games=None, node_capacity=None, log=print, dist=None):
        """n_games: concurrent self-play games over ALL ranks.  dist: None = use torch.distributed when it is initialised with more
        than one rank; False = single process; True = go through the collectives even at world size 1"""
        self.game, self.args, self.log = game, args, log
        if not hasattr(nnet, 'save_checkpoint'):                     # a bare torch module: give it the NeuralNet surface
            from . import train as _train
            engine_module = isinstance(nnet, (_train.SplendorV80Module, _train.AzulV84Module, _train.SantoriniV89Module, _train.SantoriniV78Module))
            w = NNetWrapper(game, dict(nn_version=getattr(nnet, 'version', -1), learn_rate=_get(args, 'learn_rate', 3e-3),
                                       batch_size=_get(args, 'batch_size', 512), epochs=_get(args, 'epochs', 2),
                                       q_weight=_get(args, 'q_weight', 0.5), dropout=_get(args, 'dropout', 0.0)),
                            module=None if engine_module else nnet)
            if engine_module:
                w.nnet = nnet
            nnet = w
        self.nnet = nnet
        # the competitor network (Coach.py:30).  A wrapper around a caller-supplied module (a game / architecture without an engine
        # net) cannot be rebuilt from args alone: the competitor gets a copy of the module
        self.pnet = (NNetWrapper(self.game, self.nnet.args, module=copy.deepcopy(self.nnet.nnet)) if getattr(self.nnet, '_custom', False)
                     else self.nnet.__class__(self.game, self.nnet.args))
        import torch.distributed as td
        self.dist = td if (dist is not False and td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or dist)) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.T = int(n_games or _get(args, 'n_games', 0) or max(1, min(4096, int(_get(args, 'numEps', 256)))))
        if self.T % self.world:
            raise ValueError('n_games %d is not a multiple of the world size %d' % (self.T, self.world))
        self.T_local = self.T // self.world
        self.cap = node_capacity
        self.trainExamplesHistory = []                                                  # Coach.py:32
        self.skipFirstSelfPlay = bool(getattr(nnet, 'requestKnowledgeTransfer', False))
        self.consecutive_failures = 0
        self.engine = None
        self.n_selfplay_waves = 0            # RNG epoch of the next wave of episodes; persisted with the examples (azg_state.json),
        self.iter_base = 0                   # like the number of iterations already played: a resumed run draws fresh randomness
        self._sym_stream = 0
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
        layout; as_tensors=True: device tensors (boards, pi, z, valids, q, meta) with the symmetries applied).  Collective when
        the Coach is distributed: rank 0 returns the examples of ALL ranks (it trains, Coach.py:150-215), the other ranks return none."""
        num_eps = int(_get(self.args, 'numEps', self.T))
        my_eps = episode_share(num_eps, self.T, self.world, self.rank)
        sims = int(_get(self.args, 'numMCTSSims', 800))
        # exactly numEps episodes, each played to its end (Coach.py:86-148); a new RNG epoch per call, so an iteration never
        # replays the games of the previous one (the reference draws fresh randomness every iteration)
        self.n_selfplay_waves += 1
        parts = []
        failure = None
        if my_eps > 0:
            net = self.nnet.evaluator(self.T_local)
            if self.engine is None:
                # the finished games' records leave the ring after every chunk of rounds (below), so the ring only has to hold what
                # one chunk can finish; args.max_examples overrides
                ring = int(_get(self.args, 'max_examples', 0)) or max(self.T_local, min(my_eps, 4 * self.T_local)) * int(_get(self.args, 'max_plies_hint', 192))
                self.engine = SelfPlayEngine(self.game, net, self.args, self.T_local, node_capacity=self.cap, max_examples=ring,
                                             rng_seed=int(_get(self.args, 'seed', 0)), stream0=self.rank * self.T_local)
            else:
                for grp in self.engine.groups:
                    grp.net = net
                self.engine.nnet, self.engine.graph = net, None          # new weights: the captured rounds are stale
            self.engine.start(epoch=self.n_selfplay_waves, episode_quota=my_eps)
            while True:
                self.engine.run(8 * max(8, sims))
                st = self.engine.stats()
                if st['errors']:
                    failure = ('rank %d: engine error flags %d (16 = example ring overflow: %d records dropped; raise args.max_examples)'
                               % (self.rank, st['errors'], st['examples_dropped']))
                    break
                parts.append(self.engine.drain_examples())
                if st['active'] == 0:
                    if st['games'] != my_eps:
                        failure = 'rank %d: %d games finished, %d expected' % (self.rank, st['games'], my_eps)
                    break
        # a failure on one rank must stop every rank BEFORE the collectives of _collect (the others would wait in them forever):
        # one MAX all_reduce of the failure flag, then the same exception everywhere
        if self.dist:
            flag = torch.tensor([1 if failure else 0], dtype=torch.int64)
            flag = flag if self.dist.get_backend() == 'gloo' else flag.to(self.game.device)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if int(flag.item()) and not failure:
                failure = 'self-play failed on another rank'
        if failure:
            raise RuntimeError(failure)
        ex = self._collect(parts)
        if as_tensors:
            return ex
        return formats.examples_to_iteration(ex, tuple(self.game.getBoardSize()),
                                             compress=not _get(self.args, 'no_compression', False),
                                             maxlen=int(_get(self.args, 'maxlenOfQueue', 10 ** 6)))

    def _collect(self, parts):
        """this rank's drained records (+ the other ranks': the one collective of the self-play path) -> canonical order (game
        stream, game index, ply) -> all symmetric forms of every record (Coach.py:66-69), on device"""
        from .selfplay import gather_examples
        dev = self.game.device
        S, A, P = self.game.S, self.game.A, self.game.P
        if parts:
            ex = [torch.cat([p[i] for p in parts], dim=0) for i in range(6)]
        else:
            ex = [torch.empty((0, S), dtype=torch.int8, device=dev), torch.empty((0, A), dtype=torch.float32, device=dev),
                  torch.empty((0, P), dtype=torch.float32, device=dev), torch.empty((0, A), dtype=torch.uint8, device=dev),
                  torch.empty((0, P), dtype=torch.float32, device=dev), torch.empty((0, 4), dtype=torch.int32, device=dev)]
        if self.dist:
            # rank 0 alone consumes the records (history, training: Coach.py:150-215): gather to root -- the count all_gather, then one
            # grouped send / receive of exactly the finished records; the other ranks only advance the symmetry stream counter
            info = {}
            ex = gather_examples(ex, dst=0, info=info)
            self.last_gather = info
            if self.rank != 0:
                self._sym_stream += int(sum(info.get('counts', [])))
                return tuple(ex)
        boards, pi, z, valids, q, meta = ex
        if boards.shape[0] == 0:
            return tuple(ex)
        m = meta.to(torch.int64)
        order = torch.argsort(m[:, 2], stable=True)                       # ply, then game index, then stream (stable sorts)
        order = order[torch.argsort(m[order, 1], stable=True)]
        order = order[torch.argsort(m[order, 0], stable=True)]
        boards, pi, z, valids, q, meta = [t[order].contiguous() for t in (boards, pi, z, valids, q, meta)]
        # games with random symmetries (The Little Prince) draw record t from stream (game.rng_seed, 2^42 + records so far + t)
        ob, op, ov, cnt = self.game.symmetries_batch(boards, pi, valids, stream0=(1 << 42) + self._sym_stream)
        self._sym_stream += int(boards.shape[0])
        K = ob.shape[1]
        keep = (torch.arange(K, device=cnt.device)[None, :] < cnt[:, None]).reshape(-1)
        rep = torch.repeat_interleave(torch.arange(boards.shape[0], device=cnt.device), cnt.to(torch.int64))
        return (ob.reshape(-1, ob.shape[2])[keep], op.reshape(-1, op.shape[2])[keep], z[rep], ov.reshape(-1, ov.shape[2])[keep], q[rep], meta[rep])

    # ---- weights across ranks ----
    def _broadcast_weights(self, module, src=0):
        """rank `src`'s parameters and buffers -> every rank, as one flat tensor per dtype (a few hundred KB .. 2 MB)"""
        if not self.dist:
            return
        sd = module.state_dict()
        gloo = self.dist.get_backend() == 'gloo'
        by_dtype = {}
        for k, v in sd.items():
            by_dtype.setdefault(v.dtype, []).append(k)
        for dt, keys in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
            flat = torch.cat([sd[k].detach().reshape(-1) for k in keys])
            buf = flat.cpu() if gloo else flat.to(self.game.device)
            self.dist.broadcast(buf, src=src)
            off = 0
            for k in keys:
                n = sd[k].numel()
                sd[k].copy_(buf[off:off + n].reshape(sd[k].shape).to(sd[k].device))
                off += n

    def _all_sum(self, values):
        if not self.dist:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        t = t if self.dist.get_backend() == 'gloo' else t.to(self.game.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    # ---- Coach.learn (:150-215) ----
    def learn(self):
        a = self.args
        ckpt = _get(a, 'checkpoint', './checkpoint')
        lead = self.rank == 0                                            # rank 0 keeps the example history, trains, writes the files
        if lead:
            os.makedirs(ckpt, exist_ok=True)
        seed = _get(a, 'seed', None)
        self._sync_resume_state()
        for i in range(1, int(_get(a, 'numIters', 1)) + 1):
            it = self.iter_base + i
            if not self.skipFirstSelfPlay or i > 1:
                if lead:
                    it_examples = self.executeEpisodes()
                    if len(it_examples) == int(_get(a, 'maxlenOfQueue', 10 ** 6)):
                        self.log('saturation of elements in iterationTrainExamples, think about decreasing numEps or increasing maxlenOfQueue')
                    self.trainExamplesHistory.append(it_examples)
                else:
                    self.executeEpisodes(as_tensors=True)                 # takes part in the gather; the records stay on rank 0
            if _get(a, 'profile', False):
                return self.results
            extra = {k: v for k, v in _vars(a).items() if isinstance(v, (int, float, bool, str, list, tuple))}
            # the competitor = the net before training (Coach.py:188-189 goes through temp.pt; rank 0 writes that file, every rank
            # copies the weights in memory)
            if lead:
                if len(self.trainExamplesHistory) > int(_get(a, 'numItersHistory', 5)):
                    self.trainExamplesHistory.pop(0)
                self.saveTrainExamples(iterations_done=it)                                                # :180
                self.nnet.save_checkpoint(folder=ckpt, filename='temp.pt', additional_keys=extra)         # :188
            self._copy_weights(self.nnet, self.pnet)
            n_examples = 0
            if lead:
                train_examples = [e for h in self.trainExamplesHistory for e in h]
                (random.Random(int(seed) * 1000003 + it) if seed is not None else random).shuffle(train_examples)     # :185
                n_examples = len(train_examples)
                self.nnet.train(train_examples, log=self.log, seed=None if seed is None else int(seed) + it)
            self._broadcast_weights(self.nnet.nnet)
            self.nnet._eval = None
            n_arena = int(_get(a, 'arenaCompare', 30))
            # the arena games are dealt out by game index (seats and random streams are functions of the index); a fresh block of
            # RNG streams per iteration (boards and chance outcomes of the arena games)
            first, cnt = split_range(n_arena, self.world, self.rank)
            tally = [0, 0, 0]
            if cnt > 0:
                arena = BatchedArena(self.game, self.nnet.evaluator(cnt), self.pnet.evaluator(cnt), a, n_parallel=cnt,
                                     node_capacity=self.cap, stream0=(1 << 32) * it, temp_for_game=self.temp_for_game, first_game_index=first)
                tally = list(arena.playGames(cnt, first_game_index=first))
                for m in arena.mcts:
                    m.forest.close()
            nwins, pwins, draws = self._all_sum(tally)
            accepted = (pwins + nwins) > 0 and float(nwins) / (pwins + nwins) >= float(_get(a, 'updateThreshold', 0.6))
            self.results.append(dict(iteration=it, examples=n_examples, nwins=nwins, pwins=pwins, draws=draws, accepted=accepted))
            if not accepted:
                self.consecutive_failures += 1
                if lead:
                    self.log('Iter #%d - new vs previous: %d-%d  (%d draws) --> REJECTED (%d)' % (it, nwins, pwins, draws, self.consecutive_failures))
                if self.consecutive_failures >= int(_get(a, 'stop_after_N_fail', 10 ** 9)) and i < int(_get(a, 'numIters', 1)):
                    if lead:
                        self.log('Exceeded threshold number of consecutive fails, stopping process')                 # :204-206
                    if _get(a, 'exit_on_fail', True):
                        sys.exit()
                    return self.results
                self._copy_weights(self.pnet, self.nnet)                                                          # :207 (== loading temp.pt)
            else:
                if lead:
                    self.log('Iter #%d - new vs previous: %d-%d  (%d draws) --> ACCEPTED' % (it, nwins, pwins, draws))
                    self.nnet.save_checkpoint(folder=ckpt, filename=self.getCheckpointFile(it), additional_keys=extra)
                    self.nnet.save_checkpoint(folder=ckpt, filename='best.pt', additional_keys=extra)
                self.consecutive_failures = 0
        return self.results

    def _sync_resume_state(self):
        """rank 0's resume state (loadTrainExamples: RNG epoch of the next self-play wave, iterations already played, symmetry stream,
        skipFirstSelfPlay) -> every rank.  The checkpoint folder may be visible to rank 0 only; ranks that disagreed on these would
        draw different random streams for the shards of one iteration and break "W ranks reproduce one rank"."""
        if not self.dist:
            return
        t = torch.tensor([self.n_selfplay_waves, self.iter_base, self._sym_stream, int(self.skipFirstSelfPlay)], dtype=torch.int64)
        t = t if self.dist.get_backend() == 'gloo' else t.to(self.game.device)
        self.dist.broadcast(t, src=0)
        self.n_selfplay_waves, self.iter_base, self._sym_stream, skip = [int(x) for x in t.tolist()]
        self.skipFirstSelfPlay = bool(skip)

    @staticmethod
    def _copy_weights(src, dst):
        """dst's module <- src's module (the in-memory form of save_checkpoint('temp.pt') + load_checkpoint('temp.pt'))"""
        if dst.nnet is None or type(dst.nnet) is not type(src.nnet):
            dst.nnet = copy.deepcopy(src.nnet)
        else:
            dst.nnet.load_state_dict(src.nnet.state_dict(), strict=True)
        dst._eval = None

    def getCheckpointFile(self, iteration):
        return 'checkpoint_' + str(iteration) + '.pt'

    def saveTrainExamples(self, iterations_done=None):                                                # :220-226
        folder = _get(self.args, 'checkpoint', './checkpoint')
        os.makedirs(folder, exist_ok=True)
        formats.save_train_examples(os.path.join(folder, 'checkpoint.examples'), self.trainExamplesHistory)
        # engine-side state next to the reference's file: where the random streams of a resumed run continue (the reference draws
        # fresh OS randomness on every start; a seeded engine would replay iteration 1's boards, chance outcomes and noise)
        with open(os.path.join(folder, 'azg_state.json'), 'w') as f:
            json.dump(dict(selfplay_waves=self.n_selfplay_waves, iterations_done=int(iterations_done if iterations_done is not None else self.iter_base),
                           sym_stream=self._sym_stream), f)

    def loadTrainExamples(self):                                                                      # :228-263
        model_file = _get(self.args, 'load_folder_file', None)
        model_file = os.path.join(*model_file) if isinstance(model_file, (list, tuple)) else model_file
        path = os.path.join(os.path.dirname(model_file or ''), 'checkpoint.examples')
        if self.rank != 0:
            return          # rank 0 keeps the history and trains; its resume state reaches the other ranks in learn() (_sync_resume_state)
        if not os.path.isfile(path):
            self.log('File "%s" with trainExamples not found!' % path)
            return
        with open(path, 'rb') as f:
            self.trainExamplesHistory = pickle.load(f)
        state = os.path.join(os.path.dirname(path), 'azg_state.json')
        if os.path.isfile(state):                                        # continue the random streams where the saved run stopped
            with open(state) as f:
                st = json.load(f)
            self.n_selfplay_waves, self.iter_base = int(st.get('selfplay_waves', 0)), int(st.get('iterations_done', 0))
            self._sym_stream = int(st.get('sym_stream', 0))
        else:                                                            # a history written by the reference: start behind it
            self.n_selfplay_waves = self.iter_base = len(self.trainExamplesHistory)
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
