"""Coach.executeEpisodes on the engine (Coach.py:86-148): T games per GPU, device-resident episode state machines, one
batched NeuralNet.predict per lock-step round.  The reference time-slices N game threads on one core around a lock ring
to build inference batches of N (Coach.py:117-144); here every round is
    select (HIP; the expansion + backup of the previous round's leaves rides in its prologue) -> predict_batch (MFMA net kernel)
with no host decision in the loop, so it is captured once in a HIP graph and replayed.  selfplay_advance only has work
for a tree once per search (numMCTSSims rounds), so it is launched every `advance_every` rounds: a tree whose search is
finished (or whose fresh root waits for its Dirichlet noise) sits out up to advance_every-1 rounds -- (advance_every-1) /
numMCTSSims of its time -- in exchange for one launch less per round; per-tree results do not depend on the cadence.
`work_budget` caps the work of a tree in one select launch (a launch lasts as long as its slowest tree): near the end of a
game most simulations end on terminal nodes and would otherwise all run inside one launch (measured: 0.65 ms rounds).
Whole-game sweep at 4096 x 800 sims (env-steps/s), round 1: budget 0 -> 21.3 k, 96 -> 30.9 k, 48 -> 33.8 k, 32 -> 35.1 k, 20 -> 37.2 k;
advance every 4 / 8 / 16 / 32 rounds -> 32.5 / 33.8 / 34.4 / 34.7 k.  Round 3 (tools/archive/sweep_budget_games.sh): Splendor 10 / 48 -> 68.1 k
against 63.0 k for 20 / 16; Azul and Santorini keep 20 / 16 -- the defaults below are per game family.

GROUPS (round 4, measured, NOT the default).  Both kernels of a round are latency chains -- a select launch lasts as long as the
slowest of its trees (median wave 20 us, last wave 33 us at 4096 trees), a net launch as long as one workgroup's 16 leaves -- and
neither can share a CU with the other (a net workgroup takes 504 of a SIMD's 512 VGPRs and 156 of the CU's 160 KB of LDS).  `groups`
> 1 splits the games into independent forests, each with its own stream, leaf / pi / v buffers and captured graph of K fused rounds;
nothing synchronises the groups.  The streams ask for a CU mask of one XCD each (azg_stream_create_xcd) -- but on this MI355X (one
partition over 8 XCDs) the hardware deals a queue's workgroups round-robin over ALL XCDs and the mask is not honoured
(tools/archive/dbg_placement.py: every masked stream ran on the 256 CUs of all 8 XCDs), so the groups share every CU: a net workgroup then
waits for a whole CU to drain while the other groups' descents keep landing on it.  Driver flags, one MI355X: 1 group 72.6 k
env-steps/s, 2 groups 72.3 k, 4 groups 40.5 k, 8 groups 27.4 k (profiles/archive/r04_round_structure.md).  Per-tree results do not depend on
the grouping (global game streams stream0 + index; tested)."""
import ctypes as C

import torch

from .forest import Forest


class _Group:
    def __init__(self, forest, net, shape, device_noise):
        self.f, self.net, self.shape, self.device_noise = forest, net, shape, device_noise
        st = getattr(net, 'pi', None)
        static = torch.is_tensor(st) and tuple(st.shape) == (forest.T, forest.A) and st.dtype == torch.float32
        self.pi = st if static else torch.zeros((forest.T, forest.A), dtype=torch.float32, device=forest.device)
        self.v = net.v if static else torch.zeros((forest.T, forest.P), dtype=torch.float32, device=forest.device)
        self.stream = None          # torch stream of this group's pipeline (None: the caller's current stream)
        self.graph = None
        self.async_cfg = {}         # n_net / n_sel / batch_wait_ticks of the asynchronous pipeline (defaults of the C-ABI when empty)

    def select(self):
        self.f.select(device_noise=self.device_noise)

    def predict_expand(self):
        f = self.f
        pi, v = self.net.predict_batch(f.leaf_states.view(self.shape), f.leaf_valid)
        f.expand_backup(pi, v, device_noise=self.device_noise)

    # fused form: the expansion of round r rides on the descent launch of round r + 1 (azg_forest_select_fused)
    def select_fused(self):
        self.f.select_fused(self.pi, self.v, device_noise=bool(self.device_noise))

    def predict_into_buffers(self):
        f = self.f
        pi, v = self.net.predict_batch(f.leaf_states.view(self.shape), f.leaf_valid)
        if pi.data_ptr() != self.pi.data_ptr():          # nets without static output buffers: keep the addresses stable
            self.pi.copy_(pi); self.v.copy_(v)

    def rounds(self, n, fused, percu, advance=True):
        """n lock-step rounds of this group's trees on the current stream, then (advance) one selfplay_advance.  percu: ONE launch of
        the per-CU round kernel (azg_forest_rounds_v80_h2: 16 trees + their 16 leaves per workgroup, no launch boundary between rounds);
        percu == 'async': ONE launch of the asynchronous pipeline (azg_forest_async_rounds_v80_h2: n descent / forward pairs per tree)"""
        if percu == 'async':
            # (no selfplay_advance: in the pipeline a tree whose search is finished advances at once, on the wave that found it finished)
            self.f.async_rounds_v80(self.net, self.pi, self.v, n, device_noise=bool(self.device_noise), **self.async_cfg)
            return
        if percu:
            self.f.rounds_v80(self.net, self.pi, self.v, n, device_noise=bool(self.device_noise))
            if advance:
                self.f.selfplay_advance()
            return
        for k in range(n):
            self.round(fused, advance and k == n - 1)

    def round(self, fused, advance=True):
        """one lock-step round of this group's trees on the current stream"""
        if fused:
            self.select_fused()                       # expansion of the previous round's leaves + this round's descent
            if advance:
                self.f.selfplay_advance()
            self.predict_into_buffers()
        else:
            self.select()
            self.predict_expand()
            if advance:
                self.f.selfplay_advance()


class SelfPlayEngine:
    def __init__(self, game, nnet, args, n_games, node_capacity=None, max_examples=None, rng_seed=0, stream0=0,
                 use_graph=True, dirichlet=None, level_budget=0, groups=1, advance_every=None, work_budget=None, fused=True,
                 pin_xcd=None, percu=None, gc_high_water_pct=None, async_pipe=None, async_cfg=None, deterministic=None):
        self.game, self.args = game, args
        get = (lambda k, d: args.get(k, d)) if isinstance(args, dict) else (lambda k, d: getattr(args, k, d))
        sims = int(get('numMCTSSims', 800))
        # nodes stay until the root's round passes theirs: measured up to ~11 plies' worth of simulations in Splendor
        cap = node_capacity or max(1024, 16 * sims + 512)
        assert n_games % groups == 0
        self.T, self.G = n_games, groups
        self.fused = bool(fused)
        # defaults per game family, from whole-game sweeps at 4096 games x 800 sims (tools/archive/sweep_budget_games.sh; the timings repeat to
        # 0.1 % since k_select stopped reading the dispatch packet): Splendor 2p / 4p prefer short launches and a rarer advance (work
        # budget 10, every 48 rounds: +8 % / +3 % over 20 / 16), Azul and Santorini the opposite (20 / 16: 42.5 k vs 35.5 k, 21.6 k vs 20.9 k)
        from . import _lib
        import os
        splendor = getattr(game, 'GAME_ID', None) == _lib.SPLENDOR
        Tg = n_games // groups
        nets = nnet if isinstance(nnet, (list, tuple)) else [nnet] + [nnet.clone_buffers() for _ in range(groups - 1)]
        # the per-CU round kernel and the asynchronous pipeline exist for Splendor 2 players + the V80 net on its f16 x 2 kernel with static
        # output buffers
        A_game = _lib.game_info(game.GAME_ID, game.variant)[1]
        can = (splendor and int(getattr(game, 'variant', 0) or 2) == 2 and self.fused and
               all(getattr(n, 'h2', False) and getattr(n, 'fused_net', False) and hasattr(n, 'net_ptrs_h2') and
                   torch.is_tensor(getattr(n, 'pi', None)) and tuple(n.pi.shape) == (Tg, A_game) for n in nets))
        if percu is None:
            # opt-in (AZG_PERCU=1 / percu=True): measured on one MI355X at 4096 x 800 it is the slower form of the round (65.1 k against
            # 72.5 k env-steps/s, DESIGN.md 3.6: the 16-wave / 128-VGPR net phase costs 40 us against 32 for the 12-wave kernel and the
            # launch ends with its slowest workgroup, which eats what the per-CU boundary gains on the descents)
            percu = can and os.environ.get('AZG_PERCU', '0') == '1'
        elif percu and not can:
            raise ValueError('percu=True needs Splendor 2 players and SplendorV80Hip(h2=True) evaluators with max_batch == games per group')
        # the asynchronous tree pipeline (csrc/azg_async.hip.h): persistent descent + net workgroups, no launch-wide boundary between the
        # descents and the forwards; round 5: 100 k env-steps/s against 79 k for the two-kernel rounds at 4096 x 800.  Same preconditions
        # as the per-CU kernel, one group.  AZG_ASYNC=0 / async_pipe=False: the two-kernel rounds.
        # ... and for Santorini without gods + the V89 net on its f16 x 2 kernel (SantoriniV89Hip(h2=True)) with static output buffers
        can_c5 = (getattr(game, 'GAME_ID', None) == _lib.SANTORINI and int(getattr(game, 'variant', 0) or 0) == 1 and self.fused and
                  hasattr(_lib.lib(), 'azg_forest_async_rounds_conv5_h2') and
                  all(type(n).__name__ == 'SantoriniV89Hip' and getattr(n, 'h2', False) and torch.is_tensor(getattr(n, 'pi', None)) and
                      tuple(n.pi.shape) == (Tg, A_game) for n in nets))
        # ... and for Splendor 3 / 4 players and Azul + their MobileNet-1d nets on the f16 x 2 one-launch kernel (MobileNet1dHip(h2=True, fused))
        gid, var = getattr(game, 'GAME_ID', None), int(getattr(game, 'variant', 0) or 0)
        can_mb = (self.fused and hasattr(_lib.lib(), 'azg_forest_async_rounds_mb1d_h2') and
                  ((gid == _lib.SPLENDOR and var in (3, 4)) or gid == _lib.AZUL) and
                  all(type(n).__name__ == 'MobileNet1dHip' and getattr(n, 'h2', False) and getattr(n, 'fused', False) and
                      getattr(n, 'geometry', None) == ({3: 1, 4: 2}.get(var) if gid == _lib.SPLENDOR else 3) and
                      torch.is_tensor(getattr(n, 'pi', None)) and tuple(n.pi.shape) == (Tg, A_game) for n in nets))
        # ... and, for the parity tests, with the integer hash-net as the evaluator (tests/hashnet.py HashNetPipeline: async_hashnet = True)
        can_hash = (self.fused and hasattr(_lib.lib(), 'azg_forest_async_rounds_hashnet') and all(getattr(n, 'async_hashnet', False) for n in nets) and
                    ((gid == _lib.SPLENDOR and var in (0, 2, 3, 4)) or (gid == _lib.SANTORINI and var == 1) or gid == _lib.AZUL))
        any_pipe = can or can_c5 or can_mb or can_hash
        if async_pipe is None:
            async_pipe = any_pipe and groups == 1 and not percu and os.environ.get('AZG_ASYNC', '1') == '1'
        elif async_pipe and not (any_pipe and groups == 1):
            raise ValueError('async_pipe=True needs Splendor 2 players + SplendorV80Hip(h2=True), Santorini no-gods + SantoriniV89Hip(h2=True) or Splendor 3 / 4 '
                             'players / Azul + MobileNet1dHip(h2=True) evaluators with max_batch == n_games (or the tests\' hash-net), groups == 1')
        self.async_pipe = bool(async_pipe)
        self.adaptive = False
        if work_budget is None:
            # (the pipeline: a call that runs into the budget is followed by the next at once, the budget only bounds how long the trees
            # take to notice the end of a launch: 20 measured best of 0 / 10 / 20 / 40)
            work_budget = (20 if self.async_pipe else 10) if splendor else 20
        # cadence of the advance launch: idle share (K-1)/numMCTSSims kept to a few per cent
        self.K = max(1, min(48, sims // 16) if splendor else min(16, sims // 50)) if advance_every is None else int(advance_every)
        self.work_budget = int(work_budget)
        alpha = float(get('dirichletAlpha', 0.0)) if dirichlet is None else float(dirichlet)
        # Coach passes dirichlet_noise=(dirichletAlpha != 0) (Coach.py:31,96).  The Gamma variates of
        # rng.dirichlet([alpha]*n_valid) (MCTS.py:187-192) are drawn on device by the engine's own sampler.
        self.groups = []
        for g in range(groups):
            f = Forest(game.GAME_ID, game.variant, Tg, args, node_capacity=cap,
                       max_examples=(max_examples or n_games * 64) // groups, rng_seed=rng_seed,
                       stream0=stream0 + g * Tg, device=str(game.device), level_budget=level_budget, work_budget=work_budget,
                       gc_high_water_pct=70 if gc_high_water_pct is None else gc_high_water_pct)
            self.groups.append(_Group(f, nets[g], (Tg,) + tuple(f.board_shape()), 'deferred' if alpha != 0.0 else False))
        self.forest = self.groups[0].f
        self.nnet = nets[0]
        self.use_graph = use_graph
        self.rounds = 0
        self.percu = bool(percu)
        if self.async_pipe:
            self.percu = 'async'
            cfg = dict(async_cfg or {})
            for k, e in (('n_net', 'AZG_ASYNC_NNET'), ('n_sel', 'AZG_ASYNC_NSEL'), ('batch_wait_ticks', 'AZG_ASYNC_WAIT'), ('shared_budget', 'AZG_ASYNC_SHARED')):
                if k not in cfg and os.environ.get(e):
                    cfg[k] = int(os.environ[e])
            # run(rounds) = rounds x n_games calls for the trees TOGETHER (no tree waits for the slowest at the end of a launch);
            # async_cfg=dict(shared_budget=False): exactly `rounds` calls per tree (results a function of `rounds` alone: the parity tests)
            # WHAT run(rounds) LEAVES BEHIND is then a function of (seed, rounds) only up to scheduling: every game is still played move for
            # move as its RNG stream dictates (per-game results are identical, tests/test_gpu_selfplay.py
            # test_async_pipeline_shared_budget_plays_the_same_games), but how far each game has got after a given number of rounds -- and
            # therefore WHICH games have ended and delivered examples -- depends on GPU timing.  With an episode quota the set of examples
            # is the same either way.  deterministic=True (or AZG_DETERMINISTIC=1): the per-tree budget, as in the two-kernel rounds.
            if deterministic is None:
                deterministic = os.environ.get('AZG_DETERMINISTIC', '0') == '1'
            cfg.setdefault('shared_budget', not deterministic)
            self.deterministic = not cfg['shared_budget']
            self._early_seen = 0
            self.groups[0].async_cfg = cfg
            # adaptive split (opt-in, AZG_ASYNC_ADAPT=1; only with the work-sharing budget -- with per-tree budgets the results must not
            # depend on anything measured): round 5 measured it and it LOSES to the fixed half-and-half split -- every chunk boundary is a
            # launch tail + a counter read-back (chunks of 100 / 200 / 800 rounds: 84.6 / 88.9 / 92.1 k env-steps/s against 93.1 k fixed)
            self.adaptive = bool(cfg['shared_budget']) and 'n_net' not in cfg and os.environ.get('AZG_ASYNC_ADAPT', '0') == '1'
            self.adapt_chunk = int(os.environ.get('AZG_ASYNC_CHUNK', '200'))
            self._adapt_last = None
            self.split_log = []
            self.use_graph = False          # two launches per K rounds: nothing to amortise
        # one stream per pipeline; pinned to an XCD (or an equal share of the 8 XCDs) unless pin_xcd=False
        # (default off: on this MI355X a stream's CU mask is not honoured, profiles/archive/r04_placement.txt, and azg_stream_create_xcd refuses
        # devices whose CU count is not a multiple of 8)
        self.pin_xcd = False if pin_xcd is None else bool(pin_xcd)
        self._raw_streams = []
        if groups > 1:
            n_xcd = 8
            for g, grp in enumerate(self.groups):
                if self.pin_xcd:
                    per = max(1, n_xcd // groups)
                    first = (g * per) % n_xcd
                    h = C.c_void_p()
                    _lib.check(_lib.lib().azg_stream_create_xcd(first, per, C.byref(h)))
                    self._raw_streams.append(h)
                    grp.stream = torch.cuda.ExternalStream(h.value, device=game.device)
                    grp.xcds = (first, per)
                else:
                    grp.stream = torch.cuda.Stream(device=game.device)

    @property
    def graph(self):
        """the captured rounds of the first group (None = nothing captured / stale)"""
        return self.groups[0].graph

    @graph.setter
    def graph(self, value):
        assert value is None
        for grp in self.groups:
            grp.graph = None

    def close(self):
        for grp in self.groups:
            grp.graph = None
            grp.f.close()
        torch.cuda.synchronize()
        from . import _lib
        for grp in self.groups:
            grp.stream = None
        for h in self._raw_streams:
            _lib.lib().azg_stream_destroy(h)
        self._raw_streams = []

    def start(self, init_boards=None, epoch=0, episode_quota=0):
        """epoch: re-keys every random stream (Coach passes the iteration number, so that successive iterations do not
        replay the same games); episode_quota = numEps of Coach.executeEpisodes: exactly that many games are played, each to
        its end (0 = trees restart forever; the caller decides when to stop)"""
        Tg = self.T // self.G
        # the kernel deals a quota out as quota / T (+1 for the first quota % T trees); global stream t of the whole engine must play what
        # it would play in ONE forest of T trees, whatever the grouping: group g gets the sum over its streams.  Every share is computed
        # before anything is launched; a group whose share is empty plays nothing (quota -1 of the C-ABI: every tree idle).
        a, b = divmod(int(episode_quota), self.T)
        shares = [(a * Tg + max(0, min(Tg, b - g * Tg))) if episode_quota else 0 for g in range(self.G)]
        for g, grp in enumerate(self.groups):
            q = shares[g] if episode_quota == 0 or shares[g] > 0 else -1
            grp.f.selfplay_start(None if init_boards is None else init_boards[g * Tg:(g + 1) * Tg], epoch=epoch, episode_quota=q)
        if (epoch, episode_quota) != getattr(self, '_epoch_quota', (0, 0)):
            self.graph = None                # seed and quota are kernel arguments: the captured rounds are stale
        self._epoch_quota = (epoch, episode_quota)
        torch.cuda.synchronize()

    def set_search_params(self, numMCTSSims, prob_fullMCTS):
        """args.numMCTSSims / args.prob_fullMCTS for the searches that begin from now on (MCTS.py:58-59 reads them per move);
        the captured rounds are stale afterwards"""
        for grp in self.groups:
            grp.f.set_search_params(numMCTSSims, prob_fullMCTS)
        self.graph = None

    def _round(self, advance=True):
        """one eager round of every group (each on its own stream when there are several)"""
        if self.G == 1:
            self.groups[0].rounds(1, self.fused, self.percu, advance)
            return
        cur = torch.cuda.current_stream()
        for grp in self.groups:
            grp.stream.wait_stream(cur)
            with torch.cuda.stream(grp.stream):
                grp.rounds(1, self.fused, self.percu, advance)
        for grp in self.groups:
            cur.wait_stream(grp.stream)

    def capture(self):
        """Capture K rounds of every group in a HIP graph of its own (after a few eager warm-up rounds on a side stream)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                for grp in self.groups:
                    grp.rounds(1, self.fused, self.percu)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for grp in self.groups:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                grp.rounds(self.K, self.fused, self.percu, advance=True)
            grp.graph = g
        torch.cuda.synchronize()
        self.rounds += 3 + self.K

    def run(self, rounds):
        if self.async_pipe:
            # launches of the pipeline: `rounds` calls per tree (a call = what a round of the two-kernel form does for the tree); moves,
            # example records, clean-ups and root noise happen inside, tree by tree
            grp = self.groups[0]
            if not self.adaptive:
                grp.rounds(rounds, self.fused, self.percu)
                if getattr(self, 'deterministic', False):
                    # Per-tree budgets: the entry point follows its launch with a catch-up launch (what a launch that ended early -- the
                    # platform froze a workgroup, csrc/azg_async.hip.h "Recovery" -- left over); should that one have ended early as well,
                    # further catch-up launches (rounds = 0) follow here, so that "every tree has had exactly its calls" holds when run()
                    # returns.  (Reads the pipeline's counters: a device synchronisation per run(), in this mode only.)
                    for _ in range(16):
                        to = self.forest.async_profile(reset=False)['timeouts']
                        n = to['select'] + to['net']
                        if n == self._early_seen:
                            break
                        self._early_seen = n
                        grp.rounds(0, self.fused, self.percu)
            else:
                # ADAPTIVE CU SPLIT: which side is the bottleneck changes with the phase of the games (late game: many simulations end on
                # terminal nodes and need no forward -> the descents are; opening / middle game: every simulation needs one -> the net is,
                # measured 107 k vs 73 k plies/s with a fixed split).  The launch is cut into chunks; after each one the busy shares of the
                # two kinds of workgroup (counters of the kernels) move 8 CUs to the busier side.
                done = 0
                while done < rounds:
                    k = min(self.adapt_chunk, rounds - done)
                    grp.rounds(k, self.fused, self.percu)
                    done += k
                    if done < rounds or k == self.adapt_chunk:
                        self._adapt_split()
            self.rounds += rounds
            return
        if self.use_graph and self.graph is None:
            self.capture()
        done = 0
        if self.graph is not None:
            n = rounds // self.K
            if self.G == 1:
                for _ in range(n):
                    self.groups[0].graph.replay()
            elif n:
                # every group's replays go to its own (XCD-pinned) stream: independent pipelines, joined at the end of the call
                cur = torch.cuda.current_stream()
                for grp in self.groups:
                    grp.stream.wait_stream(cur)
                for _ in range(n):
                    for grp in self.groups:
                        with torch.cuda.stream(grp.stream):
                            grp.graph.replay()
                for grp in self.groups:
                    cur.wait_stream(grp.stream)
            done = rounds - rounds % self.K
        for _ in range(rounds - done):               # remainder (or no graph): eager rounds, advance every round
            self._round()
        self.rounds += rounds

    def _adapt_split(self):
        """move 8 CUs to the busier kind of workgroup (asynchronous pipeline, see run)"""
        f, cfg = self.forest, self.groups[0].async_cfg
        o = f.async_counters()
        last, self._adapt_last = self._adapt_last, o
        if last is None or o[9] < last[9]:               # (first chunk, or the counters were reset in between)
            return
        d = [a - b for a, b in zip(o, last)]
        n_sel, n_net = int(o[12]), int(o[13])
        if d[11] <= 0 or d[1] + d[2] <= 0:
            return
        net_busy, sel_busy = d[5] / d[11], d[1] / (d[1] + d[2])
        n_cu, step = n_sel + n_net, 8
        lo_sel = max(16, -(-self.T // 128))
        if net_busy - sel_busy > 0.06 and n_sel - step >= lo_sel:
            n_net, n_sel = n_net + step, n_sel - step
        elif sel_busy - net_busy > 0.06 and n_net - step >= 32:
            n_net, n_sel = n_net - step, n_sel + step
        cfg['n_net'], cfg['n_sel'] = n_net, n_sel
        self.split_log.append((n_net, round(net_busy, 3), round(sel_busy, 3)))
        assert n_net + n_sel == n_cu

    def stats(self):
        tot = None
        for grp in self.groups:
            s = grp.f.stats()
            if tot is None:
                tot = dict(s)
            else:
                for k, v in s.items():
                    tot[k] = (max(tot[k], v) if k in ('max_nodes', 'max_live_after_gc') else tot[k] | v if k == 'errors' else
                              [a + b for a, b in zip(tot[k], v)] if k == 'cyc_seg' else tot[k] + v)
        return tot

    @property
    def device_bytes(self):
        return sum(grp.f.device_bytes for grp in self.groups)

    def drain_examples(self, symmetries=False):
        """-> (boards int8[n,S], pi f32[n,A], z f32[n,P], valids u8[n,A], q f32[n,P], meta i32[n,4]) of finished games
        (Coach.py:76-82 record layout).  symmetries=True expands every record into all its Game.getSymmetries forms on
        device, in the order Coach.py:66-69 appends them (all forms of a ply, identity first); z, q and meta are repeated."""
        parts = [grp.f.drain_examples() for grp in self.groups]
        ex = tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(6))
        if not symmetries or ex[0].shape[0] == 0:
            return ex
        boards, pi, z, valids, q, meta = ex
        # games with random symmetries (The Little Prince) draw record t from stream (game.rng_seed, 2^42 + records drained so far + t)
        self._sym_stream = getattr(self, '_sym_stream', 0)
        ob, op, ov, cnt = self.game.symmetries_batch(boards.contiguous(), pi.contiguous(), valids.contiguous(),
                                                     stream0=(1 << 42) + self._sym_stream)
        self._sym_stream += int(boards.shape[0])
        K = ob.shape[1]
        keep = (torch.arange(K, device=cnt.device)[None, :] < cnt[:, None]).reshape(-1)
        rep = torch.repeat_interleave(torch.arange(boards.shape[0], device=cnt.device), cnt.to(torch.int64))
        return (ob.reshape(-1, ob.shape[2])[keep], op.reshape(-1, op.shape[2])[keep], z[rep],
                ov.reshape(-1, ov.shape[2])[keep], q[rep], meta[rep])


def pack_records(tensors):
    """(boards, pi, z, valids, q, meta, ...) with a common first dimension n -> ONE uint8 tensor [n, row_bytes]: every record is one
    byte row (fields at 4-byte aligned offsets), so that the episode-end exchange is a single collective on a single buffer"""
    n = int(tensors[0].shape[0])
    layout, off = [], 0
    for t in tensors:
        nb = int(t[0].numel() if n else torch.empty((1,) + tuple(t.shape[1:]), dtype=t.dtype).numel()) * t.element_size()
        layout.append((off, nb, t.dtype, tuple(t.shape[1:])))
        off += (nb + 3) // 4 * 4
    rows = torch.zeros((n, off), dtype=torch.uint8, device=tensors[0].device)
    for t, (o, nb, _, _) in zip(tensors, layout):
        if n:
            rows[:, o:o + nb] = t.contiguous().view(n, -1).view(torch.uint8).view(n, nb)
    return rows, layout


def unpack_records(rows, layout):
    n = int(rows.shape[0])
    return [rows[:, o:o + nb].contiguous().view(dt).view((n,) + shape) for (o, nb, dt, shape) in layout]


def gather_examples(tensors, group=None, dst=None, info=None):
    """Multi-GPU episode-end gather of variable-length example records over RCCL (torch.distributed 'nccl' backend on ROCm).  With
    games sharded embarrassingly this is the ONLY exchange on the path: an all_gather of the record COUNTS (8 bytes
    per rank), then ONE data collective on the records packed as byte rows (pack_records):
      dst = r    -> only rank r receives (Coach.py:150-215 trains in one place): one grouped send / receive of exactly count[k]
                    rows from every rank k (torch.distributed.batch_isend_irecv = one ncclGroup on RCCL) -- no padding, nothing
                    replicated; the other ranks get empty tensors back;
      dst = None -> every rank receives every record: one all_gather of the rows padded to the largest count.
    info (a dict, optional) receives world, counts per rank, bytes received by this rank and the wall time of the exchange."""
    import os
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tensors
    if dist.get_world_size(group) == 1 and not os.environ.get('AZG_FORCE_DIST'):     # AZG_FORCE_DIST: run the collectives at world 1 too
        return tensors
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = tensors[0].device
    if dist.get_backend(group) == 'gloo' and dev.type != 'cpu':       # (CPU collectives: the world-2 tests on one GPU)
        out = gather_examples([t.cpu() for t in tensors], group, dst, info)
        return [t.to(dev) for t in out]
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = torch.tensor([tensors[0].shape[0]], dtype=torch.int64, device=dev)
    cnt = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnt, n, group=group) if dist.get_backend(group) != 'gloo' else dist.all_gather(list(cnt.split(1)), n, group=group)
    counts = [int(c) for c in cnt.tolist()]
    rows, layout = pack_records(tensors)
    rb = int(rows.shape[1])
    if dst is None:
        m = max(counts)
        pad = torch.zeros((m, rb), dtype=torch.uint8, device=dev)
        pad[:rows.shape[0]] = rows
        allr = torch.empty((world, m, rb), dtype=torch.uint8, device=dev)
        if dist.get_backend(group) != 'gloo':
            dist.all_gather_into_tensor(allr.view(world * m, rb), pad, group=group)
        else:
            dist.all_gather(list(allr.unbind(0)), pad, group=group)
        got = torch.cat([allr[k, :c] for k, c in enumerate(counts)], dim=0)
    elif rank == dst:
        got = torch.empty((sum(counts), rb), dtype=torch.uint8, device=dev)
        offs = [sum(counts[:k]) for k in range(world)]
        got[offs[rank]:offs[rank] + counts[rank]] = rows
        ops = [dist.P2POp(dist.irecv, got[offs[k]:offs[k] + counts[k]], k if group is None else dist.get_global_rank(group, k), group)
               for k in range(world) if k != rank and counts[k] > 0]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
    else:
        got = rows[:0]
        if counts[rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, rows, dst if group is None else dist.get_global_rank(group, dst), group)]):
                w.wait()
    out = unpack_records(got, layout)
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)
    if info is not None:
        info.update(world=world, counts=counts, row_bytes=rb, bytes_received=int(got.shape[0]) * rb if (dst is None or rank == dst) else 0,
                    ms=(time.perf_counter() - t0) * 1e3, mode='all_gather (padded)' if dst is None else 'grouped send/recv to rank %d' % dst,
                    backend=dist.get_backend(group))
    return out
