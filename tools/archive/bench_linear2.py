import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from azg_amd.nnet import SplendorV80Hip
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=4096)
blk = net.trunk
for M in (16, 64, 1024, 7168, 28672):
    for _ in range(50):
        net._linear(net.x1, net.C, blk.pWe, blk.be, net.h, net.E, M, net.C, net.E, act=1)       # <11,0,4>
        net._linear(net.h, net.E, blk.pWp, blk.bp, net.x2, net.C, M, net.E, net.C)               # <4,0,11>
        net._linear(net.x0, net.C, net.pW0, net.b0, net.x1, net.C, M, net.C, net.C)              # <4,0,4>
torch.cuda.synchronize()
