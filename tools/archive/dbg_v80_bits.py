"""bitwise comparison of the V80 forward variants: writes pi / v of AZG_V80_WAVES (env) to argv[1]; with two files, compares them"""
import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    print('pi equal', torch.equal(a['pi'], b['pi']), 'v equal', torch.equal(a['v'], b['v']), 'max |dpi|', float((a['pi'] - b['pi']).abs().max()), 'max |dv|', float((a['v'] - b['v']).abs().max()))
    sys.exit(0)
from azg_amd import nnet
T = 4096
torch.manual_seed(3)
w = os.path.join(R, 'tests/golden/weights_splendor2_v80.npz')
boards = torch.randint(0, 5, (T, 56, 7), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 81), device='cuda:0') < 0.5).to(torch.uint8); valids[:, -1] = 1
net = nnet.SplendorV80Hip.from_npz(w, device='cuda:0', max_batch=T, h2=True)
pi, v = net.predict_batch(boards, valids)
torch.save(dict(pi=pi.cpu(), v=v.cpu()), sys.argv[1])
