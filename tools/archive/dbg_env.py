import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'oracle')]
import numpy as np, torch
import azg_oracle as O
from azg_amd import games
for players in (2, 3, 4):
    g = games.SplendorGame(players); og = O.OracleGame(O.SPLENDOR, players)
    rng = np.random.default_rng(players); n = 64
    boards = np.stack([og.getInitBoard(og.rng(seed=5, stream=i)).reshape(-1) for i in range(n)])
    player = np.zeros(n, dtype=np.int32); bad = 0
    for ply in range(120):
        acts = np.zeros(n, dtype=np.int32)
        for i in range(n):
            v = og.getValidMoves(boards[i], int(player[i])); idx = np.flatnonzero(v); buy = idx[idx < 30]
            acts[i] = int(rng.choice(buy)) if len(buy) and rng.random() < 0.7 else int(rng.choice(idx))
        seeds = torch.full((n,), 31416 + ply, dtype=torch.int64, device=g.device)
        nb, npl = g.next_state_batch(torch.from_numpy(boards).to(g.device), torch.from_numpy(player).to(g.device), torch.from_numpy(acts).to(g.device), seeds)
        nb, npl = nb.cpu().numpy(), npl.cpu().numpy()
        for i in range(n):
            eb, ep = og.getNextState(boards[i], int(player[i]), int(acts[i]), random_seed=31416 + ply)
            eb = eb.reshape(-1)
            if not np.array_equal(nb[i], eb) and bad < 4:
                bad += 1
                d = np.flatnonzero(nb[i] != eb)
                print('P', players, 'ply', ply, 'i', i, 'player', player[i], 'act', acts[i], 'diff bytes', d[:20], 'rows', sorted(set(d // 7)), 'got', nb[i][d][:20], 'exp', eb[d][:20])
            nb[i] = eb
        boards, player = nb, npl.astype(np.int32)
    print('players', players, 'mismatches (first 4 shown):', bad)
