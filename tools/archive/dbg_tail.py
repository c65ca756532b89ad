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
e = SelfPlayEngine(g, net, a, T, node_capacity=13312, max_examples=T*160, use_graph=False)
e.start(); e.run(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
L = _lib.lib(); L.azg_debug_tree_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
f = e.forest
def snap(which=0):
    out = np.zeros(T, dtype=np.uint64); L.azg_debug_tree_cycles(f.h, which, out.ctypes.data_as(C.c_void_p)); return out.astype(np.int64)
f.enable_timing(True)
for rep in range(3):
    w0 = snap(10); s0 = snap(); l0 = snap(1); e0 = snap(2); nl0 = snap(4); ne0 = snap(5); sg0 = [snap(6 + k) for k in range(4)]
    ms0 = f.kernel_ms(0)
    e.run(1)
    w1 = snap(10); s1 = snap(); l1 = snap(1); e1 = snap(2); nl1 = snap(4); ne1 = snap(5); sg1 = [snap(6 + k) for k in range(4)]
    ms1 = f.kernel_ms(0)
    d = s1 - s0
    print('launch %d: per-tree cycles mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | levels mean %.0f max %.0f | edge mean %.0f max %.0f | select_ms %s' % (
        rep, d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max(), (l1-l0).mean(), (l1-l0).max(), (e1-e0).mean(), (e1-e0).max(), ms1))
    nl, ne = nl1 - nl0, ne1 - ne0
    print('   edges/launch histogram:', np.bincount(ne.astype(int))[:8], ' levels mean %.2f max %d' % (nl.mean(), nl.max()))
    for ne_k in range(0, 4):
        sel = ne == ne_k
        if sel.any():
            print('   edges=%d: n=%d cycles mean %.0f max %.0f, levels mean %.1f, edge-cycles mean %.0f' % (ne_k, sel.sum(), d[sel].mean(), d[sel].max(), nl[sel].mean(), (e1-e0)[sel].mean()))
    top = np.argsort(-d)[:8]
    for i in top:
        print('   slow tree %d: cycles %d levels %d edges %d levelcyc %d edgecyc %d seg(load,move,canon,probe) %s' % (i, d[i], nl[i], ne[i], (l1-l0)[i], (e1-e0)[i], [int((sg1[k]-sg0[k])[i]) for k in range(4)]))
    wt = (w1 - w0).astype(float)
    if wt.max() > 0:
        print('   wall (100 MHz) per tree: mean %.1f us max %.1f us => shader clock %.0f MHz; kernel %.1f us' % (wt.mean() / 100, wt.max() / 100, d.sum() / wt.sum() * 100, ms1[0] * 1000 / max(1, ms1[1]) if False else 0))
