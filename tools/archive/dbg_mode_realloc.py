"""does the slow / fast mode of k_select follow the PROCESS or the ALLOCATION?  Builds the bench engine several times in
one process (freeing everything in between) and prints the select time of each incarnation."""
import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R)
import torch
import bench
a = bench.argparse.Namespace(net_dtype='fp32', net='hip', groups=1, sims=800, prob_full=1.0, node_capacity=0, no_graph=False, level_budget=0,
                             work_budget=20, advance_every=0, roofline_rounds=100, traffic_json='none', preroll_plies=0)
keep = []
decoy = None
last_slow = False
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    if os.environ.get('DECOY'):
        # DECOY=GB: after a slow incarnation, occupy GB of device memory first, so that the next forest gets other physical pages
        decoy = None
        torch.cuda.empty_cache()
        if last_slow:
            decoy = torch.empty(int(float(os.environ['DECOY']) * 2 ** 30), dtype=torch.uint8, device='cuda:0')
    eng, margs, label, weights, net_kind = bench.build_engine(a, 'splendor2', 4096, 0, 'cuda:0')
    eng.game_key = 'splendor2'
    eng.start(); eng.run(1600)
    torch.cuda.synchronize()
    r = bench.measure_roofline(a, eng, 4096)
    smi = ''
    if os.environ.get('SMI'):
        import subprocess, re
        out = subprocess.run(['rocm-smi', '--showtemp', '--showpower', '--showclocks', '--showmemuse'], capture_output=True, text=True).stdout
        smi = ' | '.join(re.sub(r'GPU\[0\]\s*:\s*', '', ln).strip() for ln in out.splitlines() if 'GPU[0]' in ln)
    last_slow = r['select_ms'] > 0.053
    print('decoy' if decoy is not None else 'plain', 'incarnation', k, 'select_ms', round(r['select_ms'], 4), 'heap ptr %x' % eng.forest.leaf_states.data_ptr(), smi, flush=True)
    if os.environ.get('HOLD') and r['select_ms'] > 0.056:
        keep.append(eng)          # keep the slow allocation alive, so that the next incarnation gets other memory
        continue
    for grp in eng.groups:
        grp.f.close()
    del eng
    torch.cuda.empty_cache()
