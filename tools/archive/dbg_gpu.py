import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import azg_oracle as O
from azg_amd import games
from azg_amd.mcts import BatchedMCTS
from hashnet import HashNetTorch
class Args(dict): __getattr__ = dict.get
d = np.load(os.path.join(ROOT, 'tests/golden/mcts_splendor4_numba.npz'))
g = games.SplendorGame(4); og = O.OracleGame(O.SPLENDOR, 4)
net = HashNetTorch(4)
for i in range(len(d['case_sims'])):
    args = Args(numMCTSSims=int(d['case_sims'][i]), cpuct=float(d['case_cpuct'][i]), fpu=float(d['case_fpu'][i]), universes=int(d['case_universes'][i]), forced_playouts=bool(d['case_forced'][i]), prob_fullMCTS=1.0)
    m = BatchedMCTS(g, net, args, 1, node_capacity=1024)
    m.getActionProb(torch.from_numpy(d['case_root'][i:i+1]).cuda(), temp=1, force_full_search=True)
    rs = m.forest.root_stats()
    q = rs['Qsa'][0].cpu().numpy(); e = d['case_Qsa'][i]
    bad = np.flatnonzero(q != e)
    print(i, int(d['case_sims'][i]), 'Nsa ok', np.array_equal(rs['Nsa'][0].cpu().numpy(), d['case_Nsa'][i]), 'bad', bad[:5], [(q[b], e[b], q[b]-e[b]) for b in bad[:3]], 'Qs', float(rs['Qs'][0]), float(d['case_Qs'][i]))
    if len(bad) and int(d['case_sims'][i]) == 25:
        # compare hashnet vs oracle hashnet on all nodes of the tree
        tree = m.forest.dump_tree(0)
        for k in range(tree['n']):
            st = tree['states'][k]
            va = og.getValidMoves(st, 0)
            pi_o, v_o = O.hashnet_predict(og, st, va)
            pi_t, v_t = net.predict_batch(torch.from_numpy(st[None]).cuda(), torch.from_numpy(va[None]).cuda())
            if not np.array_equal(v_o, v_t[0].cpu().numpy()) or not np.array_equal(pi_o, pi_t[0].cpu().numpy()):
                print('  hashnet mismatch node', k, v_o, v_t[0].cpu().numpy())
        break
    m.forest.close()
