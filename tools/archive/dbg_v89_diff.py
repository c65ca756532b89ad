"""which samples / outputs of the f16 x 2 Santorini kernel differ from the f32-MFMA kernel (debugging aid)"""
import os, sys, torch, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
base = nnet.SantoriniV89.from_npz(os.path.join(R, 'tests/golden/weights_santorini1_v89.npz'), device='cuda:0')
d = np.load(os.path.join(R, 'tests/golden/netfwd_santorini1_v89.npz'))
B = 24
boards = torch.from_numpy(d['boards'][:B]).cuda().to(torch.int8).contiguous(); masks = torch.from_numpy(d['masks'][:B]).cuda().contiguous()
out = {}
for k in (False, 'h2'):
    net = nnet.SantoriniV89Hip(base, max_batch=B, split=False, h2=k == 'h2')
    pi, v = net.predict_batch(boards, masks); torch.cuda.synchronize()
    out[k] = (pi.clone().cpu().numpy(), v.clone().cpu().numpy())
dv = np.abs(out['h2'][1] - out[False][1]).max(axis=1); dp = np.abs(out['h2'][0] - out[False][0]).max(axis=1)
print('per-sample max |v| diff', np.round(dv, 6)); print('per-sample max |pi| diff', np.round(dp, 6))
