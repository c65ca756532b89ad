"""Where do launches on the XCD-pinned streams run?  (a) eager, (b) replayed from a graph captured on an ordinary stream.
Prints, per stream, the set of XCC ids and the number of distinct (XCC, SE, SH, CU) slots the workgroups reported."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from azg_amd import _lib
L = _lib.lib()
n = 2048
for per in (1, 2, 4):
    for first in range(0, 8, per):
        h = C.c_void_p(); _lib.check(L.azg_stream_create_xcd(first, per, C.byref(h)))
        st = torch.cuda.ExternalStream(h.value)
        out = torch.zeros(n, dtype=torch.int32, device='cuda')
        with torch.cuda.stream(st):
            _lib.check(L.azg_debug_placement(n, C.c_void_p(out.data_ptr()), C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        eager = (sorted(set((o & 15).tolist())), len(set(o.tolist())))
        # the same launch from a graph captured on torch's capture stream, replayed on the pinned stream
        out.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _lib.check(L.azg_debug_placement(n, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        out.zero_(); torch.cuda.synchronize()
        with torch.cuda.stream(st):
            g.replay()
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        print('XCDs [%d, %d): eager xcc ids %s (%d slots) | graph replay xcc ids %s (%d slots)' % (first, first + per, eager[0], eager[1], sorted(set((o & 15).tolist())), len(set(o.tolist()))))
        del g; torch.cuda.synchronize()
        L.azg_stream_destroy(h)
out = torch.zeros(n, dtype=torch.int32, device='cuda')
_lib.check(L.azg_debug_placement(n, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize(); o = out.cpu().numpy()
print('unpinned stream: xcc ids %s (%d slots)' % (sorted(set((o & 15).tolist())), len(set(o.tolist()))))
