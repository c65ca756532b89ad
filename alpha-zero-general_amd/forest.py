"""Thin Python handle on an `azg_forest` (include/azg.h).  torch is used only for device buffers and streams."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ForestCfg, SelfplayStats, check, lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Forest:
    """T independent search trees on one GPU.  args: the reference's `args` object (dotdict or namespace) with
    numMCTSSims, cpuct, fpu, universes, prob_fullMCTS, ratio_fullMCTS, forced_playouts, dirichletAlpha, temperature,
    tempThreshold (main.py:120-156, pit.py:49-57)."""

    def __init__(self, game_id, variant, n_trees, args, node_capacity=4096, max_examples=0, rng_seed=0, stream0=0,
                 device='cuda:0', row_capacity_bytes=0, level_budget=0, work_budget=0, gc_high_water_pct=0):
        if not torch.cuda.is_available():
            raise _lib.AzgError('no GPU visible: the engine has no CPU fallback')
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.S, self.A, self.P, self.rows, self.cols = _lib.game_info(game_id, variant)
        g = lambda k, d: getattr(args, k, d) if not isinstance(args, dict) else args.get(k, d)  # noqa: E731
        cfg = ForestCfg()
        cfg.game, cfg.variant, cfg.n_trees = game_id, variant, n_trees
        cfg.node_capacity, cfg.row_capacity_bytes = node_capacity, row_capacity_bytes
        cfg.numMCTSSims = int(g('numMCTSSims', 800))
        cfg.cpuct, cfg.fpu = float(g('cpuct', 1.0)), float(g('fpu', 0.0))
        cfg.universes = int(g('universes', 1))
        cfg.prob_fullMCTS = float(g('prob_fullMCTS', 1.0))
        cfg.ratio_fullMCTS = int(g('ratio_fullMCTS', 5))
        cfg.forced_playouts = int(bool(g('forced_playouts', True)))
        cfg.dirichletAlpha = float(g('dirichletAlpha', 0.0))
        temp = list(g('temperature', [1.0, 1.0, 1.0]))
        temp = temp + [1.0] * (3 - len(temp))        # pretrained checkpoints embed only 2 entries
        for i in range(3):
            cfg.temperature[i] = float(temp[i])
        cfg.tempThreshold = float(g('tempThreshold', 10))
        cfg.rng_seed, cfg.stream0, cfg.max_examples = rng_seed, stream0, max_examples
        cfg.level_budget = level_budget
        cfg.work_budget = work_budget
        cfg.gc_high_water_pct = int(gc_high_water_pct)
        self.cfg = cfg
        self.T = n_trees
        h = C.c_void_p()
        check(lib().azg_forest_create(C.byref(cfg), C.byref(h)))
        self.h = h
        dev = self.device
        self.leaf_states = torch.zeros((n_trees, self.S), dtype=torch.int8, device=dev)
        self.leaf_valid = torch.zeros((n_trees, self.A), dtype=torch.uint8, device=dev)
        self.needs_eval = torch.zeros((n_trees,), dtype=torch.uint8, device=dev)

    def close(self):
        if getattr(self, 'h', None):
            lib().azg_forest_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        return lib().azg_forest_device_bytes(self.h)

    def board_shape(self):
        g = self.cfg.game
        return (5, 5, 3) if g == _lib.SANTORINI else (9, 9, 4) if g == _lib.ABALONE else (66, 5, 7) if g == _lib.BOTANIK else (13, 13, self.cols) if g == _lib.AKROPOLIS else (self.rows, self.cols)

    def reset(self):
        check(lib().azg_forest_reset(self.h, _stream()))

    def begin_search(self, roots, full=None):
        roots = roots.reshape(self.T, self.S).contiguous()
        assert roots.dtype == torch.int8 and roots.is_cuda
        self._keep = (roots, full)
        check(lib().azg_forest_begin_search(self.h, _ptr(roots), _ptr(full), _stream()))

    @staticmethod
    def _stride(noise, normalised):
        if noise is None:
            return 0
        assert noise.dtype == torch.float64 and noise.is_contiguous()
        return -noise.shape[1] if normalised else noise.shape[1]

    def _noise_arg(self, noise, normalised, device_noise):
        """noise_stride of include/azg.h: device_noise True -> -1 (sampled on device right before each select),
        'deferred' -> -2 (sampled on device by the next selfplay_advance; trees wait for it)"""
        if noise is None and device_noise:
            return -2 if device_noise == 'deferred' else -1
        return self._stride(noise, normalised)

    def select(self, noise=None, normalised=False, device_noise=False):
        check(lib().azg_forest_select(self.h, _ptr(self.leaf_states), _ptr(self.leaf_valid), _ptr(self.needs_eval),
                                      _ptr(noise), self._noise_arg(noise, normalised, device_noise), _stream()))

    def select_fused(self, pi, v, device_noise=False):
        """expand_backup(pi, v) of the previous select's leaves, then the next select, in one launch (self-play)"""
        assert pi.dtype == torch.float32 and v.dtype == torch.float32 and pi.is_contiguous() and v.is_contiguous()
        assert pi.shape == (self.T, self.A) and v.shape == (self.T, self.P)
        check(lib().azg_forest_select_fused(self.h, _ptr(self.leaf_states), _ptr(self.leaf_valid), _ptr(self.needs_eval),
                                            _ptr(pi), _ptr(v), -2 if device_noise else 0, _stream()))

    def rounds_v80(self, net, pi, v, rounds, device_noise=False):
        """`rounds` x (select_fused -> V80 forward of the leaf batch) in ONE launch of the per-CU round kernel (azg_forest_rounds_v80_h2):
        Splendor 2 players with a SplendorV80Hip(h2=True) evaluator `net`; pi / v are the buffers the forward writes and the next
        expansion reads"""
        assert pi.dtype == torch.float32 and v.dtype == torch.float32 and pi.is_contiguous() and v.is_contiguous()
        assert pi.shape == (self.T, self.A) and v.shape == (self.T, self.P)
        check(lib().azg_forest_rounds_v80_h2(self.h, _ptr(self.leaf_states), _ptr(self.leaf_valid), _ptr(self.needs_eval), _ptr(pi), _ptr(v),
                                             -2 if device_noise else 0, net.net_ptrs_h2, net.descale_h2, int(rounds), _stream()))

    def async_rounds_v80(self, net, pi, v, rounds, device_noise=False, n_net=0, n_sel=0, batch_wait_ticks=-1, shared_budget=False):
        """`rounds` (descent, forward) pairs per tree as ONE launch of the asynchronous pipeline (azg_forest_async_rounds_v80_h2): persistent
        descent workgroups and persistent V80 net workgroups resident together, leaves and trees handed over through device-side queues"""
        assert pi.dtype == torch.float32 and v.dtype == torch.float32 and pi.is_contiguous() and v.is_contiguous()
        assert pi.shape == (self.T, self.A) and v.shape == (self.T, self.P)
        if getattr(net, 'async_hashnet', False):        # the tests' integer hash-net as the pipeline's evaluator (include/azg_testaids.h)
            check(lib().azg_forest_async_rounds_hashnet(self.h, _ptr(self.leaf_valid), _ptr(self.needs_eval), _ptr(pi), _ptr(v),
                                                        -2 if device_noise else 0, int(rounds), int(n_net), int(n_sel), int(batch_wait_ticks),
                                                        int(bool(shared_budget)), _stream()))
            return
        if hasattr(net, 'fused_ptrs_h2'):               # the MobileNet-1d family (MobileNet1dHip: 43 pointers + 16 descale factors + geometry)
            check(lib().azg_forest_async_rounds_mb1d_h2(self.h, int(net.geometry), _ptr(self.leaf_valid), _ptr(self.needs_eval), _ptr(pi), _ptr(v),
                                                        -2 if device_noise else 0, net.fused_ptrs_h2, net.descale_h2, int(rounds), int(n_net), int(n_sel),
                                                        int(batch_wait_ticks), int(bool(shared_budget)), _stream()))
            return
        if self.cfg.game == _lib.SANTORINI:            # the V89 net (SantoriniV89Hip: 14 pointers + the trunk's descale)
            check(lib().azg_forest_async_rounds_conv5_h2(self.h, _ptr(self.leaf_valid), _ptr(self.needs_eval), _ptr(pi), _ptr(v),
                                                         -2 if device_noise else 0, net.ptrs, float(net.descale), int(rounds), int(n_net), int(n_sel),
                                                         int(batch_wait_ticks), int(bool(shared_budget)), _stream()))
            return
        check(lib().azg_forest_async_rounds_v80_h2(self.h, _ptr(self.leaf_valid), _ptr(self.needs_eval), _ptr(pi), _ptr(v),
                                                   -2 if device_noise else 0, net.net_ptrs_h2, net.descale_h2, int(rounds), int(n_net), int(n_sel),
                                                   int(batch_wait_ticks), int(bool(shared_budget)), _stream()))

    def async_counters(self, reset=False):
        """the pipeline's raw counters (include/azg.h azg_forest_async_profile: 96 numbers, accumulated since the last reset)"""
        out = (C.c_double * 96)()
        check(lib().azg_forest_async_profile(self.h, out, int(reset)))
        return list(out)

    def async_profile(self, reset=True, since=None):
        """the pipeline's counters since the last reset (or since the raw counters `since`) as a dict (times in microseconds)"""
        o = self.async_counters(reset)
        if since is not None:
            o = [a - b if k not in (12, 13) and not 20 <= k < 32 else a for k, (a, b) in enumerate(zip(o, since))]
        d = dict(descents=o[0], batches=o[3], leaves=o[4], launches=o[9], plies_in_kernel=o[16])
        d['descent_us'] = o[1] / max(o[0], 1) / 100.0
        d['forward_us'] = o[5] / max(o[3], 1) / 100.0
        d['leaves_per_batch'] = o[4] / max(o[3], 1)
        d['leaf_wait_us'] = o[7] / max(o[4], 1) / 100.0            # queued -> claimed by a net workgroup
        d['ready_wait_us'] = o[8] / max(o[0], 1) / 100.0           # handed back by the net -> claimed by a descent wave
        d['select_wave_busy'] = o[1] / max(o[1] + o[2], 1)          # share of a descent wave's life inside select_tree
        d['net_wg_busy'] = o[5] / max(o[11], 1)                     # share of a net workgroup's life inside the forward
        d['n_sel'], d['n_net'] = int(o[12]), int(o[13])
        d['timeouts'] = dict(select=int(o[17]), net=int(o[18]), ranges_abandoned=int(o[26]), leaves_requeued=int(o[27]))        # launches that ended early in a time-out (sticky until the counters are reset)
        d['ctl'] = dict(leaf_tail=int(o[20]), leaf_head=int(o[21]), retired=int(o[22]), abort=int(o[23]), calls=int(o[24]), stop=int(o[25]))
        d['forward_cycles'] = o[14] / max(o[3], 1)                   # shader-clock cycles of a forward / a descent, and the clock they imply
        d['descent_cycles'] = o[15] / max(o[0], 1)
        d['net_cu_mhz'] = o[14] / max(o[5], 1) * 100.0
        d['select_cu_mhz'] = o[15] / max(o[1], 1) * 100.0
        d['launch_us'] = o[10] / max(o[9], 1) / max(o[12], 1) / 100.0   # a launch of the descent kernel (mean over its workgroups)
        d['select_resident_ticks'], d['net_resident_ticks'] = o[10], o[11]
        d['leaf_wait_hist_us'] = [int(x) for x in o[32:64]]
        d['ready_wait_hist_us'] = [int(x) for x in o[64:96]]
        return d

    def async_wginfo(self, reset=True):
        """per workgroup of the pipeline: (xcc, cu, se, sh, role 1 descent / 2 net, calls, shader cycles per call, us resident in the last launch)"""
        buf = (C.c_uint64 * (4 * 1024))()
        n = check(lib().azg_forest_async_wginfo(self.h, buf, 1024, int(reset)))
        out = []
        for k in range(n):
            w, role, calls, cyc = buf[4 * k], buf[4 * k + 1], buf[4 * k + 2], buf[4 * k + 3]
            out.append((int(w & 0xF), int((w >> 8) & 0xF), int((w >> 16) & 0x7), int((w >> 24) & 1), int(role & 0xFF), int(calls), cyc / max(calls, 1),
                        int(role >> 8) / 100.0))          # ... and how long the workgroup stayed in the last launch (us)
        return out

    def rounds_profile(self, reset=True):
        """(select phase us, net phase us, a wave's own descent us, rounds) per round of the per-CU round kernel since the last reset"""
        out = (C.c_double * 4)()
        check(lib().azg_forest_rounds_profile(self.h, out, int(reset)))
        return tuple(out)

    def expand_backup(self, pi, v, noise=None, normalised=False, device_noise=False):
        assert pi.dtype == torch.float32 and v.dtype == torch.float32 and pi.is_contiguous() and v.is_contiguous()
        assert pi.shape == (self.T, self.A) and v.shape == (self.T, self.P)
        check(lib().azg_forest_expand_backup(self.h, _ptr(pi), _ptr(v), _ptr(noise),
                                             self._noise_arg(noise, normalised, device_noise), _stream()))

    def active(self):
        n = C.c_int()
        check(lib().azg_forest_active(self.h, C.byref(n)))
        return n.value

    def action_probs(self, temp=1.0):
        probs = torch.empty((self.T, self.A), dtype=torch.float64, device=self.device)
        q = torch.empty((self.T, self.P), dtype=torch.float32, device=self.device)
        full = torch.empty((self.T,), dtype=torch.uint8, device=self.device)
        check(lib().azg_forest_action_probs(self.h, float(temp), _ptr(probs), _ptr(q), _ptr(full), _stream()))
        return probs, q, full

    def root_stats(self):
        d = self.device
        Ns = torch.empty((self.T,), dtype=torch.int32, device=d)
        Qs = torch.empty((self.T,), dtype=torch.float32, device=d)
        Nsa = torch.empty((self.T, self.A), dtype=torch.int32, device=d)
        Qsa = torch.empty((self.T, self.A), dtype=torch.float64, device=d)
        Ps = torch.empty((self.T, self.A), dtype=torch.float32, device=d)
        nn = torch.empty((self.T,), dtype=torch.int32, device=d)
        check(lib().azg_forest_root_stats(self.h, _ptr(Ns), _ptr(Qs), _ptr(Nsa), _ptr(Qsa), _ptr(Ps), _ptr(nn),
                                          _stream()))
        return dict(Ns=Ns, Qs=Qs, Nsa=Nsa, Qsa=Qsa, Ps=Ps, n_nodes=nn)

    def dump_tree(self, tree, max_nodes=None):
        max_nodes = max_nodes or self.cfg.node_capacity
        S, A, P = self.S, self.A, self.P
        states = np.zeros((max_nodes, S), dtype=np.int8)
        Ns = np.zeros(max_nodes, dtype=np.int32)
        Qs = np.zeros(max_nodes, dtype=np.float32)
        Es = np.zeros((max_nodes, P), dtype=np.float32)
        Nsa = np.zeros((max_nodes, A), dtype=np.int32)
        Qsa = np.zeros((max_nodes, A), dtype=np.float64)
        Ps = np.zeros((max_nodes, A), dtype=np.float32)
        hp = np.zeros(max_nodes, dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        n = check(lib().azg_forest_dump_tree(self.h, tree, max_nodes, p(states), p(Ns), p(Qs), p(Es), p(Nsa), p(Qsa),
                                             p(Ps), p(hp)))
        return dict(n=n, states=states[:n], Ns=Ns[:n], Qs=Qs[:n], Es=Es[:n], Nsa=Nsa[:n], Qsa=Qsa[:n], Ps=Ps[:n],
                    has_policy=hp[:n].astype(bool))

    def validate(self, verbose=True):
        return check(lib().azg_forest_validate(self.h, int(verbose)))

    # ---- self-play ----
    def selfplay_start(self, init_boards=None, epoch=0, episode_quota=0):
        """epoch: re-keys the forest's random streams (0 = the RNG contract's streams); episode_quota: Coach.executeEpisodes'
        numEps for this forest (every started game is played to its end, then the tree idles), 0 = restart forever"""
        ib = None if init_boards is None else init_boards.reshape(self.T, self.S).contiguous()
        self._keep_ib = ib
        check(lib().azg_selfplay_start_ex(self.h, _ptr(ib), int(epoch), int(episode_quota), _stream()))

    def selfplay_advance(self):
        check(lib().azg_selfplay_advance(self.h, _stream()))

    def stats(self):
        s = SelfplayStats()
        check(lib().azg_selfplay_stats_get(self.h, C.byref(s)))
        d = {n: (list(getattr(s, n)) if n == 'cyc_seg' else int(getattr(s, n))) for n, _ in SelfplayStats._fields_}
        d['active'] = self.active_count()
        return d

    def active_count(self):
        """trees that are still playing (searching, waiting for the net, the clean-up or their next search)"""
        n = C.c_int()
        check(lib().azg_selfplay_active(self.h, C.byref(n)))
        return n.value

    def drain_examples(self, max_records=None):
        max_records = max_records or self.cfg.max_examples
        d = self.device
        boards = torch.empty((max_records, self.S), dtype=torch.int8, device=d)
        pi = torch.empty((max_records, self.A), dtype=torch.float32, device=d)
        z = torch.empty((max_records, self.P), dtype=torch.float32, device=d)
        valids = torch.empty((max_records, self.A), dtype=torch.uint8, device=d)
        q = torch.empty((max_records, self.P), dtype=torch.float32, device=d)
        meta = torch.empty((max_records, 4), dtype=torch.int32, device=d)
        n = C.c_int()
        check(lib().azg_selfplay_drain_examples(self.h, max_records, _ptr(boards), _ptr(pi), _ptr(z), _ptr(valids),
                                                _ptr(q), _ptr(meta), C.byref(n), _stream()))
        n = n.value
        return boards[:n], pi[:n], z[:n], valids[:n], q[:n], meta[:n]

    def set_search_params(self, numMCTSSims, prob_fullMCTS):
        """args.numMCTSSims / args.prob_fullMCTS for the searches that begin from now on (the reference reads both at every
        getActionProb call, MCTS.py:58-59); captured HIP graphs of this forest are stale afterwards"""
        check(lib().azg_forest_set_search_params(self.h, int(numMCTSSims), float(prob_fullMCTS)))

    def enable_timing(self, on=True):
        check(lib().azg_forest_enable_timing(self.h, int(on)))

    def kernel_ms(self, which):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().azg_forest_last_kernel_ms(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value
