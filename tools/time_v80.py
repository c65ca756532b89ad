import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = os.path.join(R, 'tests/golden/weights_splendor2_v80.npz')
boards = torch.randint(0, 5, (T, 56, 7), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 81), device='cuda:0') < 0.5).to(torch.uint8); valids[:, -1] = 1
out = {}
for split in (False, True, False, True):
    net = nnet.SplendorV80Hip.from_npz(w, device='cuda:0', max_batch=T, split=split)
    for _ in range(5): net.predict_batch(boards, valids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): pi, v = net.predict_batch(boards, valids)
    e1.record(); torch.cuda.synchronize()
    out[split] = (pi.clone(), v.clone())
    print('k_v80_net split', split, 'us per forward of', T, ':', e0.elapsed_time(e1) * 1000 / 100)
print('max |pi| diff', float((out[True][0] - out[False][0]).abs().max()), 'max |v| diff', float((out[True][1] - out[False][1]).abs().max()))
