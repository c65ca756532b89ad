"""needs AZG_DEFINES="AZG_CYC_COUNTERS AZG_WALL_CAL": absolute start / end stamps of every k_select wave of one launch"""
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
WB = int(sys.argv[1]) if len(sys.argv) > 1 else 0
e = SelfPlayEngine(g, net, a, T, node_capacity=13312, max_examples=T*160, use_graph=False, work_budget=WB)
print('work_budget', WB)
e.start(); e.run(1500)
L = _lib.lib(); L.azg_debug_tree_cycles.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
def snap(which):
    out = np.zeros(T, dtype=np.uint64); L.azg_debug_tree_cycles(e.forest.h, which, out.ctypes.data_as(C.c_void_p)); return out.astype(np.int64)
e.forest.enable_timing(True)
for rep in range(6):
    e.run(1)
    st, en = snap(5), snap(10)
    t0 = st.min()
    s_us, e_us = (st - t0) / 100.0, (en - t0) / 100.0
    print('span %.1f us | wave start: p50 %.1f p90 %.1f max %.1f us | wave end: p50 %.1f p90 %.1f p99 %.1f max %.1f us | body mean %.1f max %.1f us | select_ms %s' % (
        e_us.max(), np.percentile(s_us, 50), np.percentile(s_us, 90), s_us.max(), np.percentile(e_us, 50), np.percentile(e_us, 90), np.percentile(e_us, 99), e_us.max(), (e_us - s_us).mean(), (e_us - s_us).max(), e.forest.kernel_ms(0)))
