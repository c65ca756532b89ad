#!/usr/bin/env python3
"""bench.py -- self-play env-steps/sec @ numMCTSSims=800, Splendor-2p, T concurrent games per MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either launch it under torch.distributed.run (one
rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE in the environment) or call it plainly: without WORLD_SIZE it re-executes
itself under torch.distributed.run with N ranks on 127.0.0.1 and rank 0 prints the one JSON line.

A "step" is one PLY WAVE of the hot path over the batch of T games per GPU: numMCTSSims lock-step rounds of
    select + expand/backup (HIP, one launch) -> NeuralNet.predict_batch (engine MFMA kernel) [-> selfplay_advance (HIP)]
i.e. one full search for every concurrent game plus the per-ply work of Coach.executeEpisode (Coach.py:61-84: policy
target, move sampling, example record, the env step Coach.py:71, end detection, restart, re-root, clean-up, root noise).
Every ply is a full numMCTSSims search (prob_fullMCTS=1).  value = env-steps/sec = plies executed in the
timed region / seconds, summed over ranks (Coach.py:71: one getNextState = one env step); the simulation-based figure
(simulations / numMCTSSims / seconds, which also counts the partial plies at the window edges) is printed next to it.

Extra objects on the JSON line: "roofline" (the select+expand+backup kernel, HIP events on the launch stream, algorithmic
bytes from live engine counters, SURVEY.md §8d formula), "cpu_baseline" (the C oracle + PyTorch-CPU net on the host
cores, bounded sample, rank 0 at N=1 only) and "secondary" (the north star's second target, Santorini no-gods, same
engine, shorter window, with its own roofline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class Args(dict):
    __getattr__ = dict.get


# MCTS / self-play settings embedded in the reference's splendor/pretrained_2players.pt
SPLENDOR2_ARGS = dict(numMCTSSims=800, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True, dirichletAlpha=0.3,
                      temperature=[1.25, 0.8, 1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
WEIGHTS = os.path.join(ROOT, 'tests', 'golden', 'weights_splendor2_v80.npz')
# other hot-path games (parity-test configs of BASELINE.json; selectable with --game, not the default bench line)
OTHER_GAMES = {
    'santorini1': dict(args=dict(numMCTSSims=800, cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True, dirichletAlpha=0.2,
                                 temperature=[1.25, 0.8, 1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0),
                       weights='weights_santorini1_v89.npz', net='SantoriniV89', label='Santorini no-gods (NB_GODS=1), V89 net'),
    'splendor4': dict(args=dict(numMCTSSims=800, cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True, dirichletAlpha=0.3,
                                temperature=[1.25, 0.8, 1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0),
                      weights='weights_splendor4_v80.npz', net='SplendorV80', label='Splendor 4p (chance nodes = 3 universes), V80 net'),
    'santorini11': dict(args=dict(numMCTSSims=800, cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True, dirichletAlpha=0.2,
                                  temperature=[1.25, 0.8, 1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0),
                        weights='weights_santorini11_v78.npz', net='SantoriniV78',
                        label='Santorini with gods (NB_GODS=11, A=1782), V78 net'),
    'azul': dict(args=dict(numMCTSSims=800, cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True, dirichletAlpha=-1,
                           temperature=[1.25, 0.8, 1.0], tempThreshold=10, ratio_fullMCTS=5, prob_fullMCTS=1.0),
                 weights='weights_azul_v84.npz', net='AzulV84', label='Azul 2p, V84 net'),
}
HBM_PEAK_GBS = 8000.0    # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = {'f16': 2500.0, 'bf16': 2500.0, 'f32': 157.3}      # dense, same guide
# algorithmic FLOPs of one leaf evaluation = 2 x MACs of the reference module
NET_MFLOP_PER_LEAF = {'splendor2': 1.042, 'azul': 0.408, 'santorini1': 18.52, 'santorini11': 13.8}
# how far the games are moved away from the opening before the warm-up (plies played with the reference's own FAST searches,
# numMCTSSims // ratio_fullMCTS simulations, MCTS.py:58-59), so that games END inside a short timed window and the episode-end
# example gather carries data; about 3/4 of a typical game
PREROLL_PLIES = {'splendor2': 40, 'splendor4': 70, 'santorini1': 14, 'santorini11': 16, 'azul': 50}


def algorithmic_bytes_per_sim(S, A, P, d, vbar, e):
    """B_sim = d*(ceil(A/8) + 12*v + 8 + S + 16) + e*(S + ceil(A/8) + 12*v + 8 + 4A + 4P)"""
    mask = (A + 7) // 8
    return d * (mask + 12.0 * vbar + 8 + S + 16) + e * (S + mask + 12.0 * vbar + 8 + 4 * A + 4 * P)


CPU_GAMES = {   # game key -> (oracle game id name, variant, torch-CPU net class, weights, MCTS settings)
    'splendor2': ('SPLENDOR', 2, 'SplendorV80', 'weights_splendor2_v80.npz', dict(cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True)),
    'santorini1': ('SANTORINI', 1, 'SantoriniV89', 'weights_santorini1_v89.npz', dict(cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True)),
}


def cpu_baseline(sims, seconds=12.0, n_par=8, game_key='splendor2'):
    """The oracle (C restatement of MCTS.py + the game's *LogicNumba, pinned against the reference) with the PyTorch-CPU net,
    one host thread, leaves batched over n_par games like --parallel-inferences 8 (Coach.py:117-144)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import azg_oracle as O
    from azg_amd import nnet as _nn
    torch.set_num_threads(1)
    gid, variant, net_cls, wfile, kw = CPU_GAMES[game_key]
    og = O.OracleGame(getattr(O, gid), variant)
    net = getattr(_nn, net_cls).from_npz(os.path.join(ROOT, 'tests', 'golden', wfile), device='cpu')
    shape = tuple(og.shape)
    trees = [O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **kw)) for _ in range(n_par)]
    boards = [og.getInitBoard(og.rng(seed=99, stream=i)) for i in range(n_par)]
    players = [0] * n_par
    rngs = [og.rng(seed=99, stream=1000 + i) for i in range(n_par)]
    for i, t in enumerate(trees):
        t.search_begin(og.getCanonicalForm(boards[i], players[i]).reshape(-1), True)
    plies = 0
    sims_done = 0
    t0 = time.perf_counter()
    lb = np.zeros((n_par, og.S), dtype=np.int8)
    lv = np.zeros((n_par, og.A), dtype=np.uint8)
    while time.perf_counter() - t0 < seconds:
        need = []
        for i, t in enumerate(trees):
            while True:
                if t.search_done():
                    probs, q, _ = t.search_end(1.0)
                    a = int(np.argmax(probs))
                    boards[i], players[i] = og.getNextState(boards[i], players[i], a, 0, rngs[i])
                    plies += 1
                    if og.getGameEnded(boards[i], players[i]).any():
                        boards[i], players[i] = og.getInitBoard(rngs[i]), 0
                        t = trees[i] = O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **kw))
                    t.search_begin(og.getCanonicalForm(boards[i], players[i]).reshape(-1), True)
                r = t.sim_begin()
                sims_done += 1
                if r == 1:
                    b, v = t.leaf()
                    lb[i], lv[i] = b, v
                    need.append(i)
                    break
        pi, vv = net.predict_batch(torch.from_numpy(lb).view((n_par,) + shape), torch.from_numpy(lv))
        pi, vv = pi.numpy(), vv.numpy()
        for i in need:
            trees[i].sim_finish(pi[i], vv[i])
    dt = time.perf_counter() - t0
    return dict(value=sims_done / sims / dt, unit='env-steps/sec', cores=1, kind='port',
                sample='%d concurrent games, %.0f s wall, %d sims (=%d plies completed) of the same %s/%d-sim '
                       'workload; C oracle tree+env, PyTorch-CPU %s net batched over %d leaves, 1 thread'
                       % (n_par, dt, sims_done, plies, game_key, sims, net_cls, n_par),
                sims_per_sec=sims_done / dt, host_cores_available=os.cpu_count())


def cpu_baseline_multi(sims, seconds, procs, game_key='splendor2'):
    """the CPU path on several host cores = `procs` independent single-thread copies of cpu_baseline (one
    process per core, 8 games each, like the reference's one-process-per-core self-play), summed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', '--sims', str(sims), '--cpu-seconds', str(seconds), '--game', game_key]
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')) for _ in range(procs)]
    rs = []
    for p in ps:
        out, _ = p.communicate(timeout=seconds * 4 + 600)
        try:
            rs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            pass
    if not rs:
        return cpu_baseline(sims, seconds, game_key=game_key)
    one = rs[0]
    return dict(value=sum(r['value'] for r in rs), unit='env-steps/sec', cores=len(rs), kind='port',
                sample='%d processes x (%s)' % (len(rs), one['sample']),
                sims_per_sec=sum(r['sims_per_sec'] for r in rs), per_core=sum(r['value'] for r in rs) / len(rs),
                host_cores_available=os.cpu_count())


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def respawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run on this node"""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:            # a rank died (torch.distributed.run has already torn the others down): say so, with a non-zero exit code
        sys.stderr.write('bench.py: the %d-rank run failed (torch.distributed.run exit code %d): no result line\n' % (n, rc))
    return rc


def build_engine(a, game_key, T, rank, dev):
    """game + engine-kernel net + SelfPlayEngine for one of the hot-path configs"""
    import torch
    from azg_amd import games
    from azg_amd import nnet as _nn
    from azg_amd.selfplay import SelfPlayEngine
    dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[a.net_dtype]
    net_kind = a.net
    if game_key == 'splendor2':
        margs = Args(SPLENDOR2_ARGS)
        game = games.SplendorGame(2, device=dev)
        label = 'Splendor 2p'
        weights = os.path.basename(WEIGHTS)
        if net_kind == 'hip':
            assert a.net_dtype == 'fp32'
            net = _nn.SplendorV80Hip.from_npz(WEIGHTS, device=dev, max_batch=T // a.groups)
        else:
            net = _nn.SplendorV80.from_npz(WEIGHTS, device=dev, dtype=dtype)
    else:
        og = OTHER_GAMES[game_key]
        margs = Args(og['args'])
        game = {'splendor4': lambda: games.SplendorGame(4, device=dev), 'santorini1': lambda: games.SantoriniGame(1, device=dev),
                'santorini11': lambda: games.SantoriniGame(11, device=dev), 'azul': lambda: games.AzulGame(device=dev)}[game_key]()
        nkw = dict(num_players=4) if game_key == 'splendor4' else {}
        net = getattr(_nn, og['net']).from_npz(os.path.join(ROOT, 'tests', 'golden', og['weights']), device=dev, dtype=dtype, **nkw)
        label, weights = og['label'], og['weights']
        if net_kind == 'hip' and og['net'] in ('SplendorV80', 'AzulV84') and a.net_dtype == 'fp32':
            net = _nn.MobileNet1dHip(net, max_batch=T // a.groups)        # the whole forward in one launch (nn_mb1d.hip.h)
        elif net_kind == 'hip' and game_key in ('santorini1', 'santorini11') and a.net_dtype == 'fp32':
            # the ResNet / the with-gods MobileNet in one launch (nn_conv5x5.hip.h)
            net = (_nn.SantoriniV89Hip if game_key == 'santorini1' else _nn.SantoriniV78Hip)(net, max_batch=T // a.groups)
        else:
            net_kind = 'torch'
    margs['numMCTSSims'] = a.sims
    margs['prob_fullMCTS'] = a.prob_full
    # nodes live until the root's age passes theirs: ~12 plies' worth of simulations in Splendor, more in the narrow, deep
    # searches of Azul / Santorini (with gods: A = 1782 makes a node ~5 KB; 14 x sims keeps 4096 trees within the 288 GB)
    # Round 4: the engine cleans a tree up once its arena is 70 % full (cfg.gc_high_water_pct), not only when another search would not
    # fit, so the arena is sized for the largest LIVE tree + one search with >= 20 % headroom (Splendor 2p: 7.5 k live nodes measured over
    # whole games -> 13 x sims; it was 16 x sims with the clean-up at exhaustion: 158.9 -> 130.5 GB for 4096 trees)
    cap = a.node_capacity or max(2048, {'splendor2': 13, 'splendor3': 24, 'splendor4': 24, 'santorini11': 14}.get(game_key, 32) * a.sims + 512)
    eng = None
    for attempt in range(3):           # the forest wants a large share of the 288 GB HBM: shrink the arena if the device has less to give
        try:
            eng = SelfPlayEngine(game, net, margs, T, node_capacity=cap, max_examples=T * 160, rng_seed=2026,
                                 stream0=rank * T, use_graph=not a.no_graph, level_budget=a.level_budget, groups=a.groups,
                                 work_budget=None if a.work_budget < 0 else a.work_budget, advance_every=a.advance_every or None,
                                 pin_xcd=None if not a.no_pin_xcd else False,
                                 async_pipe=None if getattr(a, 'async_pipe', -1) < 0 else bool(a.async_pipe))
            break
        except Exception as ex:        # azg_amd.AzgError: hipMalloc failed
            if a.node_capacity or attempt == 2:
                raise
            sys.stderr.write('forest with node_capacity %d did not fit (%s); retrying smaller\n' % (cap, ex))
            torch.cuda.empty_cache()
            cap = cap * 3 // 4
    return eng, margs, label, weights, net_kind


def measure_roofline(a, eng, T):
    """eager rounds with HIP events around the select (+ fused expand/backup) launches, on the stream they are launched on"""
    import torch
    f = eng.forest
    kprof = None
    aprof = None
    if getattr(eng, 'async_pipe', False):
        # the asynchronous pipeline: the descents run in ONE persistent kernel per `advance_every` rounds, next to the net workgroups.
        # Its launch duration (mean over its workgroups, 100 MHz wall clock inside the kernel) is the roofline's denominator; the
        # per-descent latency, the forward's latency, the queue waits and their distributions come from the same counters
        n = max(eng.K, a.roofline_rounds // eng.K * eng.K)
        f.async_profile(reset=True)
        r0 = eng.stats()
        eng.run(n)
        torch.cuda.synchronize()
        r1 = eng.stats()
        aprof = f.async_profile(reset=True)
        ms_sel, n_sel, ms_exp, n_exp = aprof['launch_us'] * 1e-3, int(aprof['launches']), 0.0, 0
    elif getattr(eng, 'percu', False):
        # the per-CU round kernel: there is no launch of the descent alone to put HIP events around -- its select phase is timed INSIDE
        # the kernel on the 100 MHz wall clock (azg_forest_rounds_profile): per round, how long a workgroup's select phase lasts (it waits
        # for the slowest of its 16 trees), averaged over the workgroups and the rounds of the same graph replays the timed region uses
        n = max(eng.K, a.roofline_rounds // eng.K * eng.K)
        f.rounds_profile(reset=True)
        r0 = eng.stats()
        eng.run(n)
        torch.cuda.synchronize()
        r1 = eng.stats()
        kprof = f.rounds_profile(reset=True)
        ms_sel, n_sel, ms_exp, n_exp = kprof[0] * 1e-3, int(kprof[3]), 0.0, 0
    else:
        r0 = eng.stats()
        f.enable_timing(True)
        for _ in range(a.roofline_rounds):
            eng._round()
        torch.cuda.synchronize()
        ms_sel, n_sel = f.kernel_ms(0)
        ms_exp, n_exp = f.kernel_ms(1)
        if not n_exp:                 # fused engine: the expansion + backup runs in the prologue of k_select, no launch of its own
            ms_exp = 0.0
        f.enable_timing(False)
        r1 = eng.stats()
    rs = r1['sims'] - r0['sims']
    if rs <= 0 or n_sel <= 0:
        return None
    d = (r1['levels'] - r0['levels']) / rs
    e = (r1['expansions'] - r0['expansions']) / rs
    vbar = (r1['sum_valid_visited'] - r0['sum_valid_visited']) / max(1, r1['levels'] - r0['levels'])
    b_sim = algorithmic_bytes_per_sim(f.S, f.A, f.P, d, vbar, e)
    rs = rs / a.groups          # the timed launches are group 0's (T/groups trees each)
    sims_per_launch = rs / n_sel
    bytes_per_launch = b_sim * sims_per_launch
    pair_ms = ms_sel + ms_exp
    achieved = bytes_per_launch / (pair_ms * 1e-3) / 1e9
    prof = None
    if os.path.exists(a.traffic_json):
        try:
            tj = json.load(open(a.traffic_json))
            if tj.get('games') == T and tj.get('sims') == a.sims and tj.get('game', 'splendor2') == eng.game_key:
                prof = dict(hbm_bytes_per_launch=tj.get('hbm_bytes_per_launch'), file=os.path.relpath(a.traffic_json, ROOT),
                            traffic_over_algorithmic=tj.get('traffic_over_algorithmic'),
                            traffic_over_algorithmic_lower_bound=tj.get('traffic_over_algorithmic_lower_bound'),
                            note='PMC FETCH_SIZE/WRITE_SIZE of an earlier rocprofv3 run of this command (separate --pmc passes, '
                                 'gfx950 corrections); NOT measured in this run')
        except Exception:
            prof = None
    # HBM bytes per launch: PMC counters cannot be read inside this run (rocprofv3 wraps the process), and a --pmc pass SERIALISES kernels,
    # which the pipeline's two concurrent persistent kernels do not survive (the descents wait for a net kernel that is never started:
    # the launch ends in its idle timeout).  The committed PMC passes therefore profile the SAME descent code as stand-alone launches
    # (AZG_ASYNC=0, tools/pmc_traffic.sh); `traffic` = their measured bytes-moved / algorithmic-bytes ratio x this launch's algorithmic bytes
    traffic = traffic_source = None
    if prof and prof.get('traffic_over_algorithmic'):
        traffic = prof['traffic_over_algorithmic'] * bytes_per_launch
        traffic_source = ('%s: (TCC FETCH_SIZE x 2 + WRITE_SIZE, gfx950 corrections) / algorithmic bytes = %.3f (an upper bound; lower bound %.3f) measured on '
                          'stand-alone launches of the same descent code, x the %.4g algorithmic bytes of this launch' %
                          (prof['file'], prof['traffic_over_algorithmic'], prof.get('traffic_over_algorithmic_lower_bound') or float('nan'), bytes_per_launch))
    extra = {}
    if aprof is not None:
        extra = dict(async_pipeline=dict((k, aprof[k]) for k in ('n_sel', 'n_net', 'leaves', 'launches', 'descents', 'plies_in_kernel', 'descent_us', 'forward_us', 'leaves_per_batch', 'leaf_wait_us',
                                                                 'ready_wait_us', 'select_wave_busy', 'net_wg_busy', 'launch_us', 'forward_cycles', 'descent_cycles', 'net_cu_mhz', 'select_cu_mhz',
                                                                 'leaf_wait_hist_us', 'ready_wait_hist_us')),
                     rounds_per_launch=eng.K,
                     timing='inside the persistent kernels on the 100 MHz wall clock (s_memrealtime): select_ms = one launch of k_async_select '
                            '(rounds_per_launch descents per tree, the waves wait for the net in between); achieved = the launch\'s algorithmic '
                            'bytes / that; descent_us = one select_tree call (expansion + backup + descent of one tree)')
    if kprof is not None:
        extra = dict(net_phase_ms=kprof[1] * 1e-3, select_wave_ms=kprof[2] * 1e-3,
                     timing='inside k_rounds_v80 on the 100 MHz wall clock (s_memrealtime), per round, averaged over the workgroups: select_ms = '
                            'the select phase of a workgroup (its slowest of 16 trees), select_wave_ms = a wave\'s own descent, net_phase_ms = the net phase')
    return dict(bound='hbm', kernels=(['k_async_select (persistent: expansion + backup + descent of the trees a workgroup owns)'] if aprof is not None else
                                      ['k_rounds_v80: select phase (expansion + backup + descent of 16 trees per workgroup)'] if kprof is not None else
                                      ['k_select', 'k_expand_backup'] if n_exp else ['k_select (expand+backup fused into its prologue)']), **extra,
                achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                traffic=traffic, traffic_source=traffic_source, traffic_from_profile=prof,
                bytes_per_sim=b_sim, sims_per_launch=sims_per_launch, bytes_per_launch=bytes_per_launch,
                select_ms=ms_sel, expand_backup_ms=ms_exp, launches=int(n_sel),
                d_levels_per_sim=d, v_valid_per_level=vbar, e_expansions_per_sim=e)


def measure_net(a, eng, T, game_key, net_kind):
    """the other kernel of a round: one NeuralNet.predict_batch over the T leaves, timed with events on the stream it is launched on
    (the net kernels go to torch's current stream); achieved = algorithmic FLOPs (2 x MACs of the reference module) / time"""
    import torch
    if game_key not in NET_MFLOP_PER_LEAF:
        return None
    grp = eng.groups[0]
    f = grp.f
    x, m = f.leaf_states.view(grp.shape), f.leaf_valid
    for _ in range(5):
        grp.net.predict_batch(x, m)
    torch.cuda.synchronize()
    # n forwards captured in a HIP graph and replayed: back-to-back kernels as in the engine's rounds.  (Launched one by one from Python
    # the figure was the HOST's launch rate whenever that was slower than the kernel -- 50 us per launch on a busy box for a 32 us kernel.)
    n = 50
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                for _ in range(n):
                    grp.net.predict_batch(x, m)
        torch.cuda.current_stream().wait_stream(side)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
    except Exception:                       # an evaluator that cannot be captured (host-side work in predict_batch): eager launches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            grp.net.predict_batch(x, m)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
    standalone_ms = ms
    rl = getattr(eng, '_last_roofline', None)
    if getattr(eng, 'percu', False) and not getattr(eng, 'async_pipe', False) and rl and rl.get('net_phase_ms'):
        ms = rl['net_phase_ms']           # the net phase of the round kernel (16 waves, timed inside the kernel); the stand-alone launch is kept beside it
    ap = (rl or {}).get('async_pipeline') if getattr(eng, 'async_pipe', False) else None
    if ap:
        # the pipeline: the forward runs inside the persistent net kernel, one batch per workgroup at a time; the kernel's rate over its
        # launch = leaves evaluated x algorithmic FLOPs per leaf / launch duration (its n_net workgroups hold n_net of the CUs)
        h2 = net_kind == 'hip' and getattr(grp.net, 'h2', False)
        dt = 'f16' if h2 else 'f32'
        flops_leaf = NET_MFLOP_PER_LEAF[game_key] * 1e6
        ach = flops_leaf * ap['leaves'] / max(ap['launch_us'] * ap['launches'], 1e-9) / 1e6
        return dict(bound='mfma', kernel='k_async_net (persistent: the forward of %s on batches of queued leaves)' % ({'splendor2': 'k_v80_net_h2<12>', 'santorini1': 'k_conv5_net<5, 162, 2, 2>'}.get(game_key, 'k_mb1d_net<Cfg, true>')),
                    achieved=ach, peak=MFMA_PEAK_TFLOPS[dt], unit='TFLOP/s', frac=ach / MFMA_PEAK_TFLOPS[dt], traffic=None, mfma_input_dtype=dt,
                    frac_of_peak_of_its_cus=ach / (MFMA_PEAK_TFLOPS[dt] * ap['n_net'] / max(1, ap['n_net'] + ap['n_sel'])),
                    note='fp32-accurate (<= 1e-5 of the reference outputs): every f32 operand is a hi + lo pair of f16 numbers, one algorithmic product = 3 '
                         'MFMAs; achieved counts ALGORITHMIC flops only, over the whole launch of the persistent kernel (its workgroups also wait for leaves)',
                    forward_us=ap['forward_us'], leaves_per_forward=ap['leaves_per_batch'], net_workgroups=ap['n_net'], net_wg_busy=ap['net_wg_busy'],
                    forward_cycles=ap.get('forward_cycles'), flops_per_leaf=flops_leaf, standalone_launch_ms=standalone_ms, standalone_leaves=T // a.groups)
    Tg = T // a.groups
    flops = NET_MFLOP_PER_LEAF[game_key] * 1e6 * Tg
    h2 = net_kind == 'hip' and getattr(grp.net, 'h2', False)
    dt = 'f16' if h2 else ('bf16' if net_kind == 'hip' and game_key in ('santorini1', 'santorini11') else 'f32')
    ach = flops / (ms * 1e-3) / 1e12
    kname = ({'splendor2': 'k_v80_net_h2', 'santorini1': 'k_conv5_net<5, 162, 2, 2> (f16 x 2 trunk)'}.get(game_key) if h2 else None) or \
        'net forward (%s, %s)' % (net_kind, type(grp.net).__name__)
    return dict(bound='mfma', kernel=kname, achieved=ach,
                peak=MFMA_PEAK_TFLOPS[dt], unit='TFLOP/s', frac=ach / MFMA_PEAK_TFLOPS[dt], traffic=None,
                mfma_input_dtype=dt, frac_of_f32_mfma_peak=ach / MFMA_PEAK_TFLOPS['f32'],
                note='fp32-accurate (<= 1e-5 of the reference outputs): every f32 operand is a hi + lo pair of 16-bit numbers, one '
                     'algorithmic product = 3 (f16 pair) or 6 (bf16 triple) MFMAs; achieved counts ALGORITHMIC flops only',
                flops_per_launch=flops, leaves_per_launch=Tg, net_ms=ms, launches=n, standalone_launch_ms=standalone_ms)


def run_workload(a, game_key, T, steps, warmup, rank, world, dev, use_dist, roofline=True, preroll_override=None):
    """warm-up ply waves, then exactly `steps` timed ply waves bracketed by barrier + synchronize; -> result dict (rank-reduced)"""
    import torch
    import torch.distributed as dist
    from azg_amd.selfplay import gather_examples
    eng, margs, label, weights, net_kind = build_engine(a, game_key, T, rank, dev)
    eng.game_key = game_key
    sims = a.sims
    eng.start()
    preroll = PREROLL_PLIES.get(game_key, 0) if a.preroll_plies < 0 else a.preroll_plies
    if preroll_override is not None:
        preroll = preroll_override
    if preroll > 0:
        fast = max(1, sims // int(margs.get('ratio_fullMCTS', 5)))
        eng.set_search_params(sims, 0.0)                     # every ply a fast search: no examples, no forced playouts
        eng.run(preroll * (fast + eng.K))
        eng.set_search_params(sims, a.prob_full)
    eng.run(warmup * sims)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    s0 = eng.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps * sims)
    ex = eng.drain_examples()
    n_local_examples = int(ex[0].shape[0])
    ginfo = {}
    if use_dist:
        # the one RCCL exchange of the path (episode-end example gather): counts all_gather + ONE grouped send / receive of the packed
        # records to rank 0, which is where Coach.learn consumes them (Coach.py:150-215)
        ex = gather_examples(list(ex), dst=0, info=ginfo)
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s1 = eng.stats()
    dt = t1 - t0
    rdev = dev if (not use_dist or dist.get_backend() == 'nccl') else 'cpu'          # (gloo: the small reductions on host tensors)
    per_rank = None
    if use_dist:
        # what every rank saw, side by side (stragglers and the memory margin beside the forest + the communicator's buffers are visible in
        # the rank-reduced line): its own wall time of the timed region, its plies, its free HBM after the run
        free_b, total_b = torch.cuda.mem_get_info()
        mine = torch.tensor([t1 - t0, float(s1['plies'] - s0['plies']), float(free_b), float(total_b)], dtype=torch.float64, device=rdev)
        allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, mine)
        per_rank = [[float(x) for x in v.tolist()] for v in allv]
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loc = torch.tensor([s1['sims'] - s0['sims'], s1['plies'] - s0['plies'], s1['errors'], s1['games'] - s0['games'],
                        n_local_examples, s1['examples_dropped']], dtype=torch.int64, device=rdev)
    if use_dist:
        # error flags are a bit mask: OR them over the ranks as a MAX per bit (RCCL has no BOR; a SUM would garble the bits)
        bits = torch.tensor([(int(loc[2]) >> b) & 1 for b in range(8)], dtype=torch.int64, device=rdev)
        dist.all_reduce(loc, op=dist.ReduceOp.SUM)
        dist.all_reduce(bits, op=dist.ReduceOp.MAX)
        loc[2] = sum(int(x) << b for b, x in enumerate(bits.tolist()))
    tot_sims, tot_plies, errs, tot_games, tot_examples, dropped = [int(x) for x in loc.tolist()]
    assert tot_plies > 0, 'no ply completed inside the timed region: raise --steps'
    res = dict(label=label, weights=weights, net_kind=net_kind, value=tot_plies / dt, value_from_sims=tot_sims / sims / dt,
               dt=dt, sims_per_sec=tot_sims / dt, plies_completed=tot_plies, games_finished=tot_games,
               examples_gathered=tot_examples if world == 1 else int(ex[0].shape[0]), examples_dropped=dropped,
               engine_errors=errs, forest_bytes_per_gpu=eng.device_bytes,
               max_live_after_gc=int(s1.get('max_live_after_gc', 0)), max_nodes_per_tree=s1['max_nodes'],
               gc_runs=s1['gc_runs'], hip_graph=eng.graph is not None, rounds_timed=steps * sims,
               ms_per_round=dt / (steps * sims) * 1e3, preroll_plies=preroll, work_budget=eng.work_budget, advance_every=eng.K, node_capacity=eng.forest.cfg.node_capacity,
               max_live_frac=int(s1.get('max_live_after_gc', 0)) / max(1, eng.forest.cfg.node_capacity))
    if use_dist:
        # proof that the collective saw every rank, and what it cost (inside the timed region): world size as the process group reports
        # it, the record count every rank contributed (from the count all_gather), the bytes rank 0 received, the wall time of the exchange
        res.update(rccl_world=dist.get_world_size(), rccl_backend=dist.get_backend(), examples_per_rank=ginfo.get('counts'),
                   gather_ms=ginfo.get('ms'), gather_bytes_received_rank0=ginfo.get('bytes_received'), gather_mode=ginfo.get('mode'),
                   gather_row_bytes=ginfo.get('row_bytes'),
                   value_per_rank=[round(v[1] / v[0], 1) for v in per_rank], ms_per_step_per_rank=[round(v[0] / steps * 1e3, 3) for v in per_rank],
                   free_hbm_bytes_per_rank=[int(v[2]) for v in per_rank], total_hbm_bytes_per_rank=[int(v[3]) for v in per_rank])
    if getattr(eng, 'async_pipe', False):
        # time-outs of the pipeline since the engine was made (sticky counters of the kernels: a launch that gave up sets error bit 128)
        try:
            res['pipeline_timeouts'] = eng.forest.async_profile(reset=False)['timeouts']
        except Exception as ex_:
            res['pipeline_timeouts'] = repr(ex_)
    res['roofline'] = measure_roofline(a, eng, T) if roofline and a.roofline_rounds > 0 else None
    eng._last_roofline = res['roofline']
    res['percu'] = bool(getattr(eng, 'percu', False)) and not getattr(eng, 'async_pipe', False)
    res['async_pipe'] = bool(getattr(eng, 'async_pipe', False))
    res['roofline_net'] = measure_net(a, eng, T, game_key, net_kind) if roofline and a.roofline_rounds > 0 else None
    del ex
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=70,
                    help='timed ply waves (one step = numMCTSSims lock-step rounds = one full search + one executed ply for every '
                         'concurrent game); the default spans one whole game per tree (~70 plies): opening, middle game, the '
                         'terminal-heavy endgame and the restart')
    ap.add_argument('--warmup', type=int, default=10, help='untimed ply waves before the timed region')
    ap.add_argument('--games', type=int, default=4096, help='concurrent games per GPU')
    ap.add_argument('--game', default='splendor2', choices=['splendor2', 'splendor4', 'santorini1', 'santorini11', 'azul'])
    ap.add_argument('--sims', type=int, default=800)
    ap.add_argument('--node-capacity', type=int, default=0)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--groups', type=int, default=1,
                    help='independent pipelines: the games are split into this many forests, each with its own stream (pinned to one XCD, or '
                         'to 8 / groups XCDs), leaf batch and captured graph; nothing synchronises them')
    ap.add_argument('--no-pin-xcd', action='store_true', help='groups > 1 on ordinary streams (no CU mask): A/B of the XCD pinning')
    ap.add_argument('--async-pipe', type=int, default=-1,
                    help='1 / 0: the asynchronous tree pipeline (persistent descent + net workgroups, csrc/azg_async.hip.h) on / off; -1 = engine default')
    ap.add_argument('--level-budget', type=int, default=0, help='max descent levels per tree per select launch (0 = unlimited)')
    ap.add_argument('--advance-every', type=int, default=0, help='rounds per selfplay_advance launch / HIP graph (0 = engine default)')
    ap.add_argument('--work-budget', type=int, default=-1, help='per-launch work cap per tree (level units), 0 = off, -1 = the engine default for the game')
    ap.add_argument('--net-dtype', default='fp32', choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--net', default='hip', choices=['hip', 'torch'], help='hip: engine MFMA kernels; torch: PyTorch-ROCm ops')
    ap.add_argument('--prob-full', type=float, default=1.0,
                    help='prob_fullMCTS: 1.0 = every ply a full search (the headline metric); 0.25 = the reference default mix of '
                         'full and numMCTSSims//5 searches (main.py), a secondary figure')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--cpu-procs', type=int, default=0,
                    help='cpu_baseline on this many host cores (independent single-thread processes, summed); 0 = min(64, host '
                         'cores): the count with the highest total on the 256-thread host -- SURVEY.md §8d asks for the host cores, not one')
    ap.add_argument('--cpu-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--roofline-rounds', type=int, default=300)
    ap.add_argument('--traffic-json', default=None, help='PMC traffic summary of an earlier profiled run (default: the newest profiles/r*_traffic.json)')
    ap.add_argument('--preroll-plies', type=int, default=-1,
                    help='plies played with fast searches before the warm-up (-1 = per-game default, about 3/4 of a game; 0 = start from '
                         'the opening): moves the games to where they end inside a short timed window, so the example gather is not empty')
    ap.add_argument('--no-secondary', action='store_true', help='skip the Santorini no-gods leg (north star\'s second target)')
    ap.add_argument('--no-sustained', action='store_true', help='skip the whole-games leg (value_whole_games)')
    ap.add_argument('--sustained-steps', type=int, default=70, help='timed ply waves of the whole-games leg')
    ap.add_argument('--secondary-steps', type=int, default=0, help='timed ply waves of the secondary leg (0 = max(3, steps // 5))')
    a = ap.parse_args()
    if a.traffic_json is None:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
        a.traffic_json = cands[-1] if cands else os.path.join(ROOT, 'profiles', 'none.json')
    if a.cpu_worker:
        print(json.dumps(cpu_baseline(a.sims, a.cpu_seconds, game_key=a.game if a.game in CPU_GAMES else 'splendor2')))
        return
    if (a.gpus > 1 or os.environ.get('AZG_BENCH_SPAWN')) and 'WORLD_SIZE' not in os.environ:   # AZG_BENCH_SPAWN: test the self-launch on one GPU
        sys.exit(respawn_ranks(a.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    use_dist = world > 1 or bool(os.environ.get('AZG_FORCE_DIST'))      # AZG_FORCE_DIST: exercise the RCCL path on one GPU
    assert world == a.gpus, 'WORLD_SIZE %d != --gpus %d' % (world, a.gpus)
    # AZG_BENCH_BACKEND=gloo: the whole multi-rank flow (spawn -> shard -> gather -> rank-reduced line) on however many GPUs there are --
    # the ranks share the devices round-robin (tests/test_gpu_bench_ranks.py runs world 2 on one GPU); default: one GPU per rank over RCCL
    backend = os.environ.get('AZG_BENCH_BACKEND', 'nccl')
    local_dev = local_rank % max(1, torch.cuda.device_count()) if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_dev)
    dev = 'cuda:%d' % local_dev
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    T = a.games
    if os.environ.get('AZG_BENCH_FAIL_RANK') and int(os.environ['AZG_BENCH_FAIL_RANK']) == rank:
        raise SystemExit('AZG_BENCH_FAIL_RANK: rank %d dies on purpose (tests/test_gpu_bench_ranks.py)' % rank)
    r = run_workload(a, a.game, T, a.steps, a.warmup, rank, world, dev, use_dist)
    search_mix = ('every ply a full search' if a.prob_full >= 1.0 else
                  'prob_fullMCTS=%g: full searches mixed with numMCTSSims//5 fast ones (reference default mix, secondary figure)' % a.prob_full)
    net_txt = ('engine kernel, fp32-accurate: f16 hi+lo split-precision MFMA on token-major tiles' if r['net_kind'] == 'hip' and a.game == 'splendor2'
               else 'engine MFMA kernels, fp32-accurate' if r['net_kind'] == 'hip' else 'PyTorch-ROCm ops')
    workload = ('Splendor 2p, numMCTSSims=%d, %d concurrent self-play games per GPU, V80 net %s (%s), args of pretrained_2players.pt '
                '(cpuct 0.8 fpu 0.0593 universes 3 forced playouts dirichlet 0.3), %s' % (a.sims, T, a.net_dtype, net_txt, search_mix)
                if a.game == 'splendor2' else
                '%s (%s), numMCTSSims=%d, %d concurrent self-play games per GPU, MCTS args of the pretrained checkpoint, %s'
                % (r['label'], net_txt, a.sims, T, search_mix))
    out = dict(metric='self-play env-steps/sec @ numMCTSSims=%d, %s' % (a.sims, 'Splendor-2p' if a.game == 'splendor2' else r['label']),
               value=r['value'], unit='env-steps/sec', n_gpus=world, steps=a.steps, warmup=a.warmup,
               ms_per_step=r['dt'] / a.steps * 1e3, higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64',
               data='synthetic (Board.init_game boards from the counter RNG; net weights: reference checkpoint converted (%s))' % r['weights'],
               config=dict(workload=workload, games_per_gpu=T,
                           step=('one ply wave = %d calls per tree on average (asynchronous pipeline: the trees share the launch\'s %d x %d calls; a call = '
                                 'what one lock-step round does for a tree)' % (a.sims, a.sims, T)) if r.get('async_pipe') else 'one ply wave = %d lock-step rounds' % a.sims,
                           parallelism='games sharded x%d by game index, no data-path collective; 1 RCCL example gather to rank 0 at episode end' % world if world > 1 else 'single GPU',
                           hip_graph=r['hip_graph']),
               groups=a.groups)
    for k in ('value_from_sims', 'sims_per_sec', 'plies_completed', 'games_finished', 'examples_gathered', 'examples_dropped',
              'engine_errors', 'forest_bytes_per_gpu', 'node_capacity', 'max_live_after_gc', 'max_live_frac', 'max_nodes_per_tree', 'gc_runs',
              'rounds_timed', 'ms_per_round', 'preroll_plies', 'work_budget', 'advance_every', 'percu', 'async_pipe'):
        out[k] = r[k]
    if 'pipeline_timeouts' in r:
        out['pipeline_timeouts'] = r['pipeline_timeouts']
    for k in ('rccl_world', 'rccl_backend', 'examples_per_rank', 'gather_ms', 'gather_bytes_received_rank0', 'gather_mode', 'gather_row_bytes',
              'value_per_rank', 'ms_per_step_per_rank', 'free_hbm_bytes_per_rank', 'total_hbm_bytes_per_rank'):
        if k in r:
            out[k] = r[k]
    out['config']['preroll'] = ('%d plies of fast searches (numMCTSSims // ratio_fullMCTS, MCTS.py:58-59) before the warm-up, untimed: games end '
                                'inside the timed window (max_nodes_per_tree hugs node_capacity by design -- the clean-up is lazy; '
                                'max_live_frac = nodes surviving a clean-up / node_capacity is the headroom figure)' % r['preroll_plies'])
    if r['roofline']:
        out['roofline'] = r['roofline']
    if r.get('roofline_net'):
        out['roofline_net'] = r['roofline_net']
    if use_dist:
        dist.barrier()
    # ---- sustained: the same workload over WHOLE games (no pre-roll: opening, middle game, endgame, restart), no roofline legs -- the
    # window above times the phase the pre-roll moved the games to; this is the rate a training run sees ----
    if a.game == 'splendor2' and not a.no_sustained and (a.preroll_plies != 0 or a.steps < 60):
        try:
            r3 = run_workload(a, a.game, T, a.sustained_steps, 10, rank, world, dev, use_dist, roofline=False, preroll_override=0)
            out['value_whole_games'] = r3['value']
            out['sustained'] = dict(value=r3['value'], unit='env-steps/sec', steps=a.sustained_steps, warmup=10, preroll_plies=0,
                                    plies_completed=r3['plies_completed'], games_finished=r3['games_finished'], engine_errors=r3['engine_errors'],
                                    ms_per_step=r3['dt'] / a.sustained_steps * 1e3,
                                    note='same engine and flags from the opening position over ~one whole game per tree')
        except Exception as ex:
            out['sustained'] = dict(error=repr(ex))
    # ---- secondary: the north star's second target (Santorini no-gods), same engine, shorter window, own roofline ----
    if a.game == 'splendor2' and not a.no_secondary:
        try:
            st = a.secondary_steps or max(3, a.steps // 5)
            r2 = run_workload(a, 'santorini1', T, st, max(1, min(a.warmup, 2)), rank, world, dev, use_dist)
            out['secondary'] = dict(metric='self-play env-steps/sec @ numMCTSSims=%d, Santorini no-gods' % a.sims, value=r2['value'],
                                    unit='env-steps/sec', steps=st, ms_per_step=r2['dt'] / st * 1e3,
                                    config=dict(workload='%s (%s), numMCTSSims=%d, %d concurrent self-play games per GPU, args of '
                                                         'santorini/pretrained.pt (cpuct 1.1 fpu 0.03 universes 0 dirichlet 0.2)'
                                                         % (r2['label'], 'engine kernel, fp32-accurate: f16 hi+lo split-precision MFMA trunk' if r2['net_kind'] == 'hip'
                                                            else 'PyTorch-ROCm ops', a.sims, T)),
                                    value_from_sims=r2['value_from_sims'], plies_completed=r2['plies_completed'],
                                    games_finished=r2['games_finished'], engine_errors=r2['engine_errors'],
                                    ms_per_round=r2['ms_per_round'], roofline=r2['roofline'], roofline_net=r2.get('roofline_net'),
                                    examples_gathered=r2['examples_gathered'], preroll_plies=r2['preroll_plies'])
        except Exception as ex:                       # the headline line must still be printed
            out['secondary'] = dict(error=repr(ex))
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.game in CPU_GAMES:
        # 64 single-thread processes: the count that gives the HIGHEST total on the 256-thread host (measured: 64 processes
        # 393 env-steps/s in total, 128 processes 297 -- they contend for memory bandwidth and the SMT siblings' units)
        procs = a.cpu_procs or max(1, min(64, os.cpu_count() or 1))
        out['cpu_baseline'] = (cpu_baseline_multi(a.sims, a.cpu_seconds, procs, a.game) if procs > 1
                               else cpu_baseline(a.sims, a.cpu_seconds, game_key=a.game))
        if a.game == 'splendor2' and isinstance(out.get('secondary'), dict) and 'value' in out['secondary']:
            out['secondary']['cpu_baseline'] = (cpu_baseline_multi(a.sims, a.cpu_seconds, procs, 'santorini1') if procs > 1
                                                else cpu_baseline(a.sims, a.cpu_seconds, game_key='santorini1'))
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
