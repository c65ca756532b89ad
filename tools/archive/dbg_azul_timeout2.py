"""in-process sequence of the full-size property tests: a 212 GB forest is closed, the next engine starts at once"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location('azg_bench', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
def one(game, sims, cap=0):
    a = bench.argparse.Namespace(net_dtype='fp32', net='hip', groups=1, sims=sims, prob_full=1.0, node_capacity=cap, no_graph=False, level_budget=0,
                                 work_budget=-1, advance_every=0, no_pin_xcd=False)
    t0 = time.time()
    eng, margs, label, weights, net_kind = bench.build_engine(a, game, 4096, 0, 'cuda:0')
    eng.start()
    torch.cuda.synchronize(); t1 = time.time()
    eng.run(3 * sims + 64)
    torch.cuda.synchronize(); t2 = time.time()
    st = eng.stats()
    p = eng.forest.async_profile(reset=False) if getattr(eng, 'async_pipe', False) else {}
    print(game, sims, 'errors', st['errors'], 'plies', st['plies'], 'build+start %.1fs run %.2fs' % (t1 - t0, t2 - t1), 'abort', (p.get('ctl') or {}).get('abort'), flush=True)
    t3 = time.time(); eng.close(); del eng; torch.cuda.empty_cache(); print('   close %.2fs' % (time.time() - t3), flush=True)
for rep in range(2):
    one('santorini11', 800); one('splendor4', 800); one('azul', 800); one('santorini1', 800); one('azul', 1600, 44000)
