"""Throughput and pipeline balance of the asynchronous pipeline ply wave by ply wave over the driver's window (pre-roll 40 plies, then 25 ply waves):
    python tools/dbg_async_phases.py [n_net n_sel [work_budget [adaptive]]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('azg_bench', os.path.join(root, 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
n_net = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_sel = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wb = int(sys.argv[3]) if len(sys.argv) > 3 else -1
adaptive = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if n_net:
    os.environ['AZG_ASYNC_NNET'], os.environ['AZG_ASYNC_NSEL'] = str(n_net), str(n_sel)
a = bench.argparse.Namespace(net_dtype='fp32', net='hip', groups=1, sims=800, prob_full=1.0, node_capacity=0, no_graph=False, level_budget=0,
                             work_budget=wb, advance_every=0, no_pin_xcd=False, async_pipe=1)
eng, margs, *_ = bench.build_engine(a, 'splendor2', 4096, 0, 'cuda:0')
if adaptive:
    eng.adaptive = True
eng.start()
fast = 800 // 5
eng.set_search_params(800, 0.0)
eng.run(40 * (fast + eng.K))
eng.set_search_params(800, 1.0)
torch.cuda.synchronize()
tot_p, tot_t = 0, 0.0
for step in range(25):
    eng.forest.async_profile(reset=True)
    s0 = eng.stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(800)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = eng.stats()
    p = eng.forest.async_profile(reset=True)
    if step >= 5:
        tot_p += s1['plies'] - s0['plies']; tot_t += dt
    print('step %2d  %6.0f plies/s  %6.0f sims/800/s  games %4d | n_net %3d n_sel %3d descent %.1f us forward %.1f us leaf_wait %.1f ready_wait %.1f sel_busy %.2f net_busy %.2f leaves/batch %.1f' % (
        step, (s1['plies'] - s0['plies']) / dt, (s1['sims'] - s0['sims']) / 800 / dt, s1['games'] - s0['games'], p['n_net'], p['n_sel'], p['descent_us'], p['forward_us'],
        p['leaf_wait_us'], p['ready_wait_us'], p['select_wave_busy'], p['net_wg_busy'], p['leaves_per_batch']))
print('steps 5..24: %.0f plies/s   errors %d' % (tot_p / tot_t, eng.stats()['errors']))
