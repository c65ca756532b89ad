"""micro-benchmark of azg_nn_linear shapes (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from azg_amd.nnet import SplendorV80Hip
net = SplendorV80Hip.from_npz(os.path.join(ROOT, 'tests/golden/weights_splendor2_v80.npz'), max_batch=4096)
B = 4096
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
blk = net.trunk
M = B * 7
print('first   ', timeit(lambda: net._linear(net.x0, net.C, net.pW0, net.b0, net.x1, net.C, M, net.C, net.C)))
print('expand  ', timeit(lambda: net._linear(net.x1, net.C, blk.pWe, blk.be, net.h, net.E, M, net.C, net.E, act=1)))
print('expand noact', timeit(lambda: net._linear(net.x1, net.C, blk.pWe, None, net.h, net.E, M, net.C, net.E, act=0)))
print('project ', timeit(lambda: net._linear(net.h, net.E, blk.pWp, blk.bp, net.x2, net.C, M, net.E, net.C, act=0, R=net.x1, ldr=net.C, rowscale=net.sc, rpg=7)))
print('project plain', timeit(lambda: net._linear(net.h, net.E, blk.pWp, blk.bp, net.x2, net.C, M, net.E, net.C)))
print('se1     ', timeit(lambda: net._linear(net.pooled, net.E, blk.pW1, blk.b1, net.se_h, 48, B, net.E, net.Q, act=1, ksplit=1)))
print('se2     ', timeit(lambda: net._linear(net.se_h, 48, blk.pW2, blk.b2, net.sc, net.E, B, 48, net.E, act=3, ksplit=1)))
print('pi1     ', timeit(lambda: net._linear(net.xh, 7 * net.C, net.pWpi1, net.bpi1, net.hid_pi, 96, B, 7 * net.C, net.A, act=1, ksplit=1)))
print('expand M/8', timeit(lambda: net._linear(net.x1, net.C, blk.pWe, blk.be, net.h, net.E, M // 8, net.C, net.E, act=1)))
x = torch.empty(28672 * 168, device='cuda'); y = torch.empty_like(x)
print('copy 19MB', timeit(lambda: y.copy_(x)))
print('empty launch', timeit(lambda: net._linear(net.x0, net.C, net.pW0, net.b0, net.x1, net.C, 16, net.C, net.C)))
