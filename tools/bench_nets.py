"""µs per forward at T leaves for every evaluation path of the MobileNet-1d nets (HIP events, 50 forwards after warm-up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from azg_amd import nnet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = os.path.join(ROOT, 'tests', 'golden')


def timed(net, boards, valids, n=50):
    for _ in range(5):
        net.predict_batch(boards, valids)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        net.predict_batch(boards, valids)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1000


for tag, mk, shape, A in [('splendor2', lambda: nnet.SplendorV80.from_npz(G + '/weights_splendor2_v80.npz', device='cuda:0'), (56, 7), 81),
                          ('splendor4', lambda: nnet.SplendorV80.from_npz(G + '/weights_splendor4_v80.npz', num_players=4, device='cuda:0'), (88, 7), 81),
                          ('azul', lambda: nnet.AzulV84.from_npz(G + '/weights_azul_v84.npz', device='cuda:0'), (23, 6), 180)]:
    base = mk()
    boards = torch.randint(0, 5, (T,) + shape, dtype=torch.int8, device='cuda:0')
    valids = (torch.rand((T, A), device='cuda:0') < 0.5).to(torch.uint8)
    valids[:, -1] = 1
    row = {'torch ops': timed(base, boards, valids.bool(), 10),
           '17 launches': timed(nnet.MobileNet1dHip(base, max_batch=T, fused=False), boards, valids),
           'k_mb1d_net f32': timed(nnet.MobileNet1dHip(base, max_batch=T, fused=True, h2=False), boards, valids),
           'k_mb1d_net h2': timed(nnet.MobileNet1dHip(base, max_batch=T, fused=True, h2=True), boards, valids)}
    if tag == 'splendor2':
        row['k_v80_net'] = timed(nnet.SplendorV80Hip.from_npz(G + '/weights_splendor2_v80.npz', max_batch=T), boards, valids)
    print(tag, 'T=%d' % T, '  '.join('%s %.1f us' % kv for kv in row.items()), flush=True)

base = nnet.SantoriniV89.from_npz(G + '/weights_santorini1_v89.npz', device='cuda:0')
boards = torch.randint(-2, 5, (T, 5, 5, 3), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 162), device='cuda:0') < 0.5).to(torch.uint8)
valids[:, 0] = 1
print('santorini1 V89 T=%d' % T, 'torch ops (MIOpen) %.1f us' % timed(base, boards, valids.bool(), 10),
      ' k_conv5_net %.1f us' % timed(nnet.SantoriniV89Hip(base, max_batch=T), boards, valids), flush=True)

base = nnet.SantoriniV78.from_npz(G + '/weights_santorini11_v78.npz', device='cuda:0')
Tg = min(T, 1024)
boards = torch.randint(-2, 5, (Tg, 5, 5, 3), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((Tg, 1782), device='cuda:0') < 0.1).to(torch.uint8)
valids[:, 0] = 1
print('santorini11 V78 T=%d' % Tg, 'torch ops (MIOpen) %.1f us' % timed(base, boards, valids.bool(), 5),
      ' k_s78_net %.1f us' % timed(nnet.SantoriniV78Hip(base, max_batch=Tg), boards, valids), flush=True)
