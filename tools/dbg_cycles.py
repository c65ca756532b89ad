"""needs the library built with AZG_DEFINES=AZG_CYC_COUNTERS python alpha-zero-general_amd/build.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from azg_amd import games
from azg_amd.nnet import SplendorV80Hip
from azg_amd.selfplay import SelfPlayEngine
class Args(dict): __getattr__ = dict.get
GAME = os.environ.get('GAME', 'splendor2')
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = os.path.join(ROOT, 'tests/golden')
if GAME == 'azul':
    from azg_amd import nnet
    a = Args(numMCTSSims=800, cpuct=0.5, fpu=0.05, universes=1, forced_playouts=True, dirichletAlpha=-1, temperature=[1.25,0.8,1.0], tempThreshold=10, ratio_fullMCTS=5, prob_fullMCTS=1.0)
    g = games.AzulGame()
    net = nnet.MobileNet1dHip(nnet.AzulV84.from_npz(G + '/weights_azul_v84.npz', device='cuda:0'), max_batch=T)
    cap = 32 * 800 + 512
elif GAME == 'santorini1':
    from azg_amd import nnet
    a = Args(numMCTSSims=800, cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True, dirichletAlpha=0.2, temperature=[1.25,0.8,1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
    g = games.SantoriniGame(1)
    net = nnet.SantoriniV89Hip(nnet.SantoriniV89.from_npz(G + '/weights_santorini1_v89.npz', device='cuda:0'), max_batch=T)
    cap = 32 * 800 + 512
elif GAME == 'santorini11':
    from azg_amd import nnet
    a = Args(numMCTSSims=800, cpuct=1.1, fpu=0.03, universes=0, forced_playouts=True, dirichletAlpha=0.2, temperature=[1.25,0.8,1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
    g = games.SantoriniGame(11)
    net = nnet.SantoriniV78Hip(nnet.SantoriniV78.from_npz(G + '/weights_santorini11_v78.npz', device='cuda:0'), max_batch=T)
    cap = 14 * 800 + 512
elif GAME == 'splendor4':
    from azg_amd import nnet
    a = Args(numMCTSSims=800, cpuct=0.8, fpu=0.1, universes=3, forced_playouts=True, dirichletAlpha=0.3, temperature=[1.25,0.8,1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
    g = games.SplendorGame(4)
    net = nnet.MobileNet1dHip(nnet.SplendorV80.from_npz(G + '/weights_splendor4_v80.npz', num_players=4, device='cuda:0'), max_batch=T)
    cap = 32 * 800 + 512
else:
    a = Args(numMCTSSims=800, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True, dirichletAlpha=0.3, temperature=[1.25,0.8,1.0], tempThreshold=6, ratio_fullMCTS=5, prob_fullMCTS=1.0)
    g = games.SplendorGame(2)
    net = SplendorV80Hip.from_npz(G + '/weights_splendor2_v80.npz', max_batch=T)
    cap = 13312
if os.environ.get('HASHNET'):           # a leaf evaluator with a tiny code footprint (does the net kernel evict k_select's code from the I-cache?)
    from hashnet import HashNetTorch
    net = HashNetTorch(g.P)
e = SelfPlayEngine(g, net, a, T, node_capacity=cap, max_examples=T*160, use_graph=False, fused=os.environ.get('FUSED', '1') == '1')
e.start(); e.run(1200)
s0 = e.stats(); e.run(300); s1 = e.stats()
seg = [s1['cyc_seg'][k]-s0['cyc_seg'][k] for k in range(4)]
d = {k: s1[k]-s0[k] for k in s0 if k != 'cyc_seg'}
n = 300 * T
print('per tree-launch cycles: select', d['cyc_select']/n, 'levels', d['cyc_levels']/n, 'edge', d['cyc_edge']/n, 'leaf', d['cyc_leaf']/n)
print('levels/launch', d['levels']/n, 'sims/launch', d['sims']/n, 'cycles per level', d['cyc_levels']/max(1,d['levels']))
print('edge split per tree-launch: prologue+load_state %.0f make_move %.0f canon+hash %.0f probe %.0f' % tuple(x/n for x in seg))
print('expansions/launch', d['expansions']/n, 'terminal', d['terminal_hits']/n)

try:
    import ctypes as C
    from azg_amd import _lib
    out = (C.c_ulonglong * 16)()
    _lib.lib().azg_debug_prolog(out, 1)
    e.run(200)
    _lib.lib().azg_debug_prolog(out, 0)
    n = max(1, out[0])
    print('prologue stamps per expansion (cycles): hdr+pi arrive %.0f | to np_sum done %.0f | entries written %.0f | backup %.0f | tail of expand_apply %.0f | to loop start %.0f'
          % tuple(out[k] / n for k in range(1, 7)))
    print('   of which end of expansion -> after the status checks: %.0f' % (out[7] / n))
except Exception as ex:
    print('no prologue stamps:', ex)
