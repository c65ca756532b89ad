"""needs AZG_DEFINES=AZG_CYC_COUNTERS build.  Per-tree k_select cycle distribution of single launches vs the launch time."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch
from azg_amd import games, _lib
from azg_amd.nnet import SplendorV80Hip
from azg_amd.selfplay import SelfPlayEngine
class Args(dict): __getattr__ = dict.get
a = Args(numMCTSSims=800, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True, dirichletAlpha=0.3, temperature=[1.25,0.8,1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
g = games.SplendorGame(2); T = 4096
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=T)
e = SelfPlayEngine(g, net, a, T, node_capacity=8512, max_examples=T*160, use_graph=False)
e.start(); e.run(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
L = _lib.lib(); L.azg_debug_tree_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
f = e.forest
def snap(which=0):
    out = np.zeros(T, dtype=np.uint64); L.azg_debug_tree_cycles(f.h, which, out.ctypes.data_as(C.c_void_p)); return out.astype(np.int64)
f.enable_timing(True)
for rep in range(4):
    s0 = snap(); l0 = snap(1); e0 = snap(2)
    ms0 = f.kernel_ms(0)
    e.run(1)
    s1 = snap(); l1 = snap(1); e1 = snap(2)
    ms1 = f.kernel_ms(0)
    d = s1 - s0
    print('launch %d: per-tree cycles mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | levels mean %.0f max %.0f | edge mean %.0f max %.0f | select_ms %s' % (
        rep, d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), (l1-l0).mean(), (l1-l0).max(), (e1-e0).mean(), (e1-e0).max(), ms1))
