"""debug build only (AZG_DEFINES=AZG_NN_PHASE_TIMES, AZG_LIB=that library): clock64 stamps of workgroup 7 / thread 0 of k_v80_net_h2"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from azg_amd import _lib
from azg_amd.nnet import SplendorV80Hip
B = 4096
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=B, h2=True)
boards = torch.randint(0, 5, (B, 56, 7), dtype=torch.int8, device='cuda')
valid = torch.ones((B, 81), dtype=torch.uint8, device='cuda')
for _ in range(5):
    net.forward(boards, valid)
torch.cuda.synchronize()
out = (C.c_longlong * 64)()
L = _lib.lib()
L.azg_nn_debug_phase_times_h2.argtypes = [C.c_void_p]
L.azg_nn_debug_phase_times_h2(out)
names = ['start', 'E expand+dw', 'S1 fc1', 'S2 fc2+H', 'P project', 'tail', 'softmax']
t0 = None
for mode in (1, 2, 3):
    t = [out[mode * 16 + k] for k in range(7)]
    if t0 is None:
        t0 = t[0]
    x = [out[mode * 16 + k] for k in range(10, 16)]
    print('   E detail: first token', x[0] - t[0], 'mfma loop', x[1] - x[0], 'depthwise', x[2] - x[1], 'to B1', t[1] - x[2], '| S1: partial', x[3] - t[1], 'wp issue', x[4] - x[3], 'barrier', x[5] - x[4], 'combine+barrier', t[2] - x[5])
    print('MODE', mode, 'begins at', t[0] - t0, ' '.join('%s=%d' % (names[k], t[k] - t[k - 1]) for k in range(1, 7) if t[k] and t[k - 1]), 'total', (max(t) - t[0]))
