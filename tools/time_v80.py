"""time the one-launch V80 forward variants on one GPU: f32 MFMAs, bf16 x 3 expand (split), fp16 hi+lo on token-major tiles (h2)"""
import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = os.path.join(R, 'tests/golden/weights_splendor2_v80.npz')
boards = torch.randint(0, 5, (T, 56, 7), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 81), device='cuda:0') < 0.5).to(torch.uint8); valids[:, -1] = 1
ref = nnet.SplendorV80.from_npz(w, device='cuda:0', dtype=torch.float64)
p64, v64 = ref.predict_batch(boards, valids.bool())
out = {}
for name, kw in (('f32', dict(split=False, h2=False)), ('split', dict(split=True, h2=False)), ('h2', dict(h2=True))) * 2:
    net = nnet.SplendorV80Hip.from_npz(w, device='cuda:0', max_batch=T, **kw)
    for _ in range(5): net.predict_batch(boards, valids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): pi, v = net.predict_batch(boards, valids)
    e1.record(); torch.cuda.synchronize()
    out[name] = (pi.clone(), v.clone())
    print('k_v80_net', name, 'us per forward of', T, ':', round(e0.elapsed_time(e1) * 1000 / 100, 2),
          ' max |pi - f64| %.3g  max |v - f64| %.3g' % (float((pi.double() - p64).abs().max()), float((v.double() - v64).abs().max())))
