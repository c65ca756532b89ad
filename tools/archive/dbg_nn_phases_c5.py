"""debug build only (AZG_DEFINES=AZG_NN_PHASE_TIMES, AZG_LIB=that library): clock64 stamps of workgroup 7 / thread 0 of k_conv5_net"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from azg_amd import _lib, nnet
B = 4096
base = nnet.SantoriniV89.from_npz(os.path.join(ROOT, 'tests/golden/weights_santorini1_v89.npz'), device='cuda:0')
net = nnet.SantoriniV89Hip(base, max_batch=B)
boards = torch.randint(-2, 5, (B, 5, 5, 3), dtype=torch.int8, device='cuda:0')
valid = torch.ones((B, 162), dtype=torch.uint8, device='cuda:0')
for _ in range(5):
    net.predict_batch(boards, valid)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
L = _lib.lib()
L.azg_nn_debug_phase_times_c5.argtypes = [C.c_void_p]
L.azg_nn_debug_phase_times_c5(out)
t = list(out)
names = {1: 'staging', 2: 'first conv', 3: 'block 0', 4: 'block 1', 5: 'block 2', 6: 'block 3', 7: 'block 4', 20: 'f32 rebuild', 21: '1x1 heads', 22: 'FCs', 23: 'softmax+v'}
prev = t[0]
for k in sorted(names):
    print('%-12s %7d cycles (100 MHz ticks x ? -- clock64)' % (names[k], t[k] - prev))
    prev = t[k]
print('total', t[23] - t[0])
print('last convolution, wave 0 (5 row tiles): prologue (tap masks, weights) %d  main loop %d  epilogue %d  barrier %d' % (t[25] - t[24], t[26] - t[25], t[27] - t[26], t[7] - t[27]))
print('first convolution, wave 0: from the phase start to its own start %d  weights arrive + split %d  tile loop %d  to the barrier %d' % (t[28] - t[1], t[29] - t[28], t[30] - t[29], t[2] - t[30]))
