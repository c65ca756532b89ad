import os, sys, torch, numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
base = nnet.SantoriniV89.from_npz(os.path.join(R, 'tests/golden/weights_santorini1_v89.npz'), device='cuda:0')
B = 4096
g = torch.Generator().manual_seed(1)
d = np.load(os.path.join(R, 'tests/golden/netfwd_santorini1_v89.npz'))
boards = torch.from_numpy(d['boards']).cuda().to(torch.int8); masks = torch.from_numpy(d['masks']).cuda()
rb = boards[torch.randint(0, boards.shape[0], (B,), generator=g)].contiguous(); rm = masks[torch.randint(0, boards.shape[0], (B,), generator=g)].contiguous(); rm[:, 0] = 1
res = {}
for split in (False, True, 'h2'):
    net = nnet.SantoriniV89Hip(base, max_batch=B, split=split is True, h2=split == 'h2')
    for _ in range(5): net.predict_batch(rb, rm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): pi, v = net.predict_batch(rb, rm)
    e1.record(); torch.cuda.synchronize()
    res[split] = (pi.clone(), v.clone())
    print('split', split, 'us per forward of', B, ':', e0.elapsed_time(e1) * 1000 / 50)
for k in (True, 'h2'):
    print(k, 'vs f32: max |pi| diff', float((res[k][0] - res[False][0]).abs().max()), 'max |v| diff', float((res[k][1] - res[False][1]).abs().max()))
