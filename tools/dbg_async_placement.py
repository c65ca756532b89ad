"""Placement study of the asynchronous pipeline: does a descent workgroup run slower when the CU next to it (same XCC / SE / SH, cu_id ^ 1:
the pair that shares an instruction cache) runs a net workgroup?  Prints mean shader cycles per call by kind of neighbour."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
spec = importlib.util.spec_from_file_location('azg_bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
a = bench.argparse.Namespace(net_dtype='fp32', net='hip', groups=1, sims=800, prob_full=1.0, node_capacity=0, no_graph=False, level_budget=0,
                             work_budget=-1, advance_every=0, no_pin_xcd=False, async_pipe=1)
eng, *_ = bench.build_engine(a, 'splendor2', 4096, 0, 'cuda:0')
eng.start()
eng.run(800)
eng.forest.async_wginfo(reset=True)
eng.run(1600)
torch.cuda.synchronize()
wi = eng.forest.async_wginfo()
by = {(x, se, sh, cu): (role, calls, cyc) for x, cu, se, sh, role, calls, cyc in wi}
print('workgroups', len(wi), 'distinct CUs', len(by))
for role in (1, 2):
    groups = {}
    for (x, se, sh, cu), (r, calls, cyc) in by.items():
        if r != role or calls == 0:
            continue
        nb = by.get((x, se, sh, cu ^ 1))
        kind = 'none' if nb is None else ('descent' if nb[0] == 1 else 'net')
        groups.setdefault(kind, []).append(cyc)
    for kind, v in sorted(groups.items()):
        print('role %s, neighbour CU (cu_id ^ 1) runs %-8s: %3d workgroups, cycles per call mean %.0f min %.0f max %.0f' % (
            'descent' if role == 1 else 'net', kind, len(v), sum(v) / len(v), min(v), max(v)))
cus = sorted(set((se, sh, cu) for (x, se, sh, cu) in by))
print('distinct (se, sh, cu) ids within an XCC:', len(cus), cus[:40])
print(eng.stats()['errors'])
