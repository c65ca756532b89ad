"""debug build only (AZG_DEFINES=AZG_NN_PHASE_TIMES, AZG_LIB=that library): clock64 stamps of workgroup 7 / thread 0 of k_s78_net_split,
block 5, second third of the expanded channels: where a pass of the with-gods trunk spends its time"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from azg_amd import _lib, nnet
B = 4096
base = nnet.SantoriniV78.from_npz(os.path.join(ROOT, 'tests/golden/weights_santorini11_v78.npz'), device='cuda:0')
net = nnet.SantoriniV78Hip(base, max_batch=B, h2=True)
boards = torch.randint(-2, 5, (B, 5, 5, 3), dtype=torch.int8, device='cuda:0')
valid = (torch.rand((B, 1782), device='cuda:0') < 0.1).to(torch.uint8); valid[:, 0] = 1
for _ in range(5):
    net.predict_batch(boards, valid)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
L = _lib.lib()
L.azg_nn_debug_phase_times_c5.argtypes = [C.c_void_p]
L.azg_nn_debug_phase_times_c5(out)
t = list(out)
names = ['expand GEMM + epilogue (wave 0)', 'barrier', 'depthwise (thread 0)', 'barrier', 'project GEMM (wave 0)', 'barrier']
for k, n in enumerate(names):
    print('%-34s %7d cycles' % (n, t[k + 1] - t[k]))
print('pass total', t[6] - t[0])
names2 = ['staging + meta FC', 'first convolution', '10 blocks', 'f32 rebuild', 'heads (1x1 convs, value FCs)']
for k, n in enumerate(names2):
    print('%-34s %7d cycles' % (n, t[9 + k] - t[8 + k]))
print('kernel total (workgroup 7, thread 0)', t[13] - t[8])
print('heads: weights staged %d | barrier %d | 1x1 convolutions %d | barrier %d | features out + fc1 %d | barrier %d | fc2 + tanh %d' %
      (t[14] - t[12], t[15] - t[14], t[16] - t[15], t[17] - t[16], t[18] - t[17], t[19] - t[18], t[13] - t[19]))
