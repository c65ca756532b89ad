"""needs an AZG_DEFINES="AZG_CYC_COUNTERS AZG_WALL_CAL" build (AZG_LIB=...).  Per-tree k_select lifetime distribution of single launches:
what a launch-wide boundary (slowest of T trees) costs against boundaries over groups of 16 / 64 / 512 trees."""
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
rows = []
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    e.run(7)
    e.run(1)
    w0 = snap(5) & 0xFFFFFFFF; w1 = snap(10) & 0xFFFFFFFF      # AZG_WALL_CAL: 100 MHz stamps of the wave's start / end
    start = w0.min()
    tot = ((w1 - start) & 0xFFFFFFFF).astype(np.float64) / 100.0    # us from the first wave's start to this wave's end
    pro = ((w0 - start) & 0xFFFFFFFF).astype(np.float64) / 100.0    # this wave's start offset
    def gmax(x, n): return x.reshape(-1, n).max(axis=1).mean()
    rows.append((tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), gmax(tot, 16), gmax(tot, 64), gmax(tot, 512), pro.mean()))
    print('launch %2d: wave end (us after launch start) mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | mean of max over groups of 16: %.1f, 64: %.1f, 512: %.1f | start offset mean %.1f' % ((rep,) + rows[-1]))
r = np.array(rows).mean(axis=0)
print('AVG: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | max16 %.1f max64 %.1f max512 %.1f | start offset %.1f' % tuple(r))
print('kernel_ms', f.kernel_ms(0))
