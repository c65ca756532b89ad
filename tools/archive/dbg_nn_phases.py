"""debug build only (hipcc -DAZG_NN_PHASE_TIMES): per-phase clock64 stamps of workgroup 7 / wave 0 of the fused V80 kernels"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from azg_amd import _lib
from azg_amd.nnet import SplendorV80Hip
B = 4096
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=B, split=os.environ.get('SPLIT', '1') == '1')
boards = torch.randint(0, 5, (B, 56, 7), dtype=torch.int8, device='cuda')
valid = torch.ones((B, 81), dtype=torch.uint8, device='cuda')
for _ in range(5):
    net.forward(boards, valid)
torch.cuda.synchronize()
out = (C.c_longlong * 64)()
L = _lib.lib()
L.azg_nn_debug_phase_times.argtypes = [C.c_void_p]
L.azg_nn_debug_phase_times(out)
names = ['start', 'pre(first layer)', 'wload+P0', 'P1 expand', 'P2 depthwise', 'P3 fc1', 'P4 fc2', 'P5 project', 'tail sync', 'tail gemms', 'softmax']
for mode in (1, 2, 3):
    t = [out[mode * 16 + k] for k in range(11)]
    print('MODE', mode, ' '.join('%s=%d' % (names[k], t[k] - t[k - 1]) for k in range(1, 11) if t[k] and t[k - 1]), 'total', (max(t) - t[0]))
