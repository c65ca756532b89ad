import os, sys, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
from azg_amd import nnet
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = nnet.SantoriniV78.from_npz(os.path.join(R, 'tests/golden/weights_santorini11_v78.npz'), device='cuda:0')
boards = torch.randint(-2, 5, (T, 5, 5, 3), dtype=torch.int8, device='cuda:0')
valids = (torch.rand((T, 1782), device='cuda:0') < 0.1).to(torch.uint8); valids[:, 0] = 1
MODE = os.environ.get("MODE", "h2")          # h2 | split | f32
net = nnet.SantoriniV78Hip(base, max_batch=T, split=MODE == "split", h2=MODE == "h2")
for _ in range(5): net.predict_batch(boards, valids)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): net.predict_batch(boards, valids)
e1.record(); torch.cuda.synchronize()
print('k_s78_net', MODE, 'us per forward of', T, ':', e0.elapsed_time(e1) * 1000 / 30)
