"""where does a long launch of the pipeline spend its time?  One launch of ROUNDS rounds (Azul, 1600 simulations by default), then per workgroup how
long it stayed, and the launch-wide busy shares.  usage: [GAME=azul SIMS=1600 ROUNDS=96000 CAP=44000] python tools/dbg_async_tail.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from azg_amd import games, nnet
from azg_amd.selfplay import SelfPlayEngine
class Args(dict): __getattr__ = dict.get
T = int(os.environ.get('T', 4096)); SIMS = int(os.environ.get('SIMS', 1600)); ROUNDS = int(os.environ.get('ROUNDS', 96000)); CAP = int(os.environ.get('CAP', 44000))
G = os.path.join(ROOT, 'tests/golden')
a = Args(numMCTSSims=SIMS, cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True, dirichletAlpha=-1, temperature=[1.25, 0.8, 1.0], tempThreshold=10, ratio_fullMCTS=5, prob_fullMCTS=1.0)
g = games.AzulGame()
net = nnet.MobileNet1dHip(nnet.AzulV84.from_npz(G + '/weights_azul_v84.npz', device='cuda:0'), max_batch=T)
e = SelfPlayEngine(g, net, a, T, node_capacity=CAP, max_examples=T * 160, use_graph=False)
e.start(); e.run(5 * SIMS); torch.cuda.synchronize()
for rep in range(int(os.environ.get('REPS', 2))):
    f = e.forest; f.async_counters(reset=True); f.async_wginfo(reset=True)
    s0 = e.stats(); t0 = time.perf_counter(); e.run(ROUNDS); torch.cuda.synchronize(); dt = time.perf_counter() - t0; s1 = e.stats()
    p = f.async_profile(reset=False); w = f.async_wginfo(reset=False)
    sel = sorted(x[7] for x in w if x[4] == 1); netw = sorted(x[7] for x in w if x[4] == 2)
    print('launch %.3f s, plies %d (%.0f /s), errors %d, busy sel %.2f net %.2f, timeouts %s' % (dt, s1['plies'] - s0['plies'], (s1['plies'] - s0['plies']) / dt, s1['errors'], p['select_wave_busy'], p['net_wg_busy'], p['timeouts']))
    q = lambda v, f_: v[int(f_ * (len(v) - 1))] / 1e6
    print('  descent workgroups stayed (s): min %.3f median %.3f p90 %.3f p99 %.3f max %.3f | net: min %.3f median %.3f max %.3f' % (q(sel, 0), q(sel, .5), q(sel, .9), q(sel, .99), q(sel, 1), q(netw, 0), q(netw, .5), q(netw, 1)))
    late = [(k, round(x[7] / 1e6, 3), x[5]) for k, x in enumerate(w) if x[4] == 1 and x[7] > 1.05 * sel[len(sel) // 2]]
    print('  descent workgroups more than 5 %% later than the median: %d %s' % (len(late), late[:12]))
