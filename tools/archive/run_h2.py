"""50 forwards of the h2 V80 kernel on 4096 leaves (target of rocprofv3 --pmc passes)"""
import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
T = 4096
w = os.path.join(R, 'tests/golden/weights_splendor2_v80.npz')
boards = torch.randint(0, 5, (T, 56, 7), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 81), device='cuda:0') < 0.5).to(torch.uint8); valids[:, -1] = 1
net = nnet.SplendorV80Hip.from_npz(w, device='cuda:0', max_batch=T, h2=True)
for _ in range(50): net.predict_batch(boards, valids)
torch.cuda.synchronize()
