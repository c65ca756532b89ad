"""outputs of the one-launch MobileNet-1d kernels (h2 and f32) for every geometry on fixed inputs -> a .pt file, or compared bit by bit with one:
   AZG_LIB=<old lib> python tools/dbg_mb1d_bits.py save /tmp/ref.pt ; python tools/dbg_mb1d_bits.py cmp /tmp/ref.pt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from azg_amd import nnet
G = os.path.join(ROOT, 'tests', 'golden')
out = {}
for tag, mk in [('splendor2_v80', lambda: nnet.SplendorV80.from_npz(G + '/weights_splendor2_v80.npz', device='cuda:0')),
                ('splendor4_v80', lambda: nnet.SplendorV80.from_npz(G + '/weights_splendor4_v80.npz', num_players=4, device='cuda:0')),
                ('azul_v84', lambda: nnet.AzulV84.from_npz(G + '/weights_azul_v84.npz', device='cuda:0')),
                ('minivilles2_v82', lambda: nnet.MobileNet1d.from_npz(G + '/weights_minivilles2_v82.npz', device='cuda:0')),
                ('tlp3_v83', lambda: nnet.MobileNet1d.from_npz(G + '/weights_tlp3_v83.npz', device='cuda:0'))]:
    d = np.load(G + '/netfwd_%s.npz' % tag)
    reps = -(-1000 // len(d['boards']))
    boards = torch.from_numpy(np.concatenate([d['boards']] * reps)[:1000]).to('cuda:0').to(torch.int8)
    masks = torch.from_numpy(np.concatenate([d['masks']] * reps)[:1000]).to('cuda:0')
    for h2 in (True, False):
        net = nnet.MobileNet1dHip(mk(), max_batch=4096, h2=h2)
        pi, v = net.predict_batch(boards, masks)
        out['%s/%d' % (tag, h2)] = (pi.clone().cpu(), v.clone().cpu())
        big = boards.repeat(5, 1, 1)[:4096].contiguous(); bm = masks.repeat(5, 1)[:4096].contiguous()
        for _ in range(3): net.predict_batch(big, bm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): net.predict_batch(big, bm)
        torch.cuda.synchronize()
        print('%-18s h2=%d  %.1f us per 4096 leaves' % (tag, h2, (time.perf_counter() - t0) / 20 * 1e6), flush=True)
if sys.argv[1] == 'save':
    torch.save(out, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    for k in out:
        print(k, 'pi identical' if torch.equal(out[k][0], ref[k][0]) else 'pi DIFFERS %.3g' % float((out[k][0] - ref[k][0]).abs().max()),
              'v identical' if torch.equal(out[k][1], ref[k][1]) else 'v DIFFERS %.3g' % float((out[k][1] - ref[k][1]).abs().max()))
