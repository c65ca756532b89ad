"""Throughput of the six f4 game plugins on one MI355X ("each to the same parity + measurement bar"): batched
self-play on the engine with the integer hash-net of SURVEY Appendix C.3 as the leaf evaluator (these games have no engine net; the
figure is the env step + tree side of the plugin, `nnet.TorchModuleEvaluator` adds the user's PyTorch-ROCm module on top).
    python tools/bench_f4.py [--games 1024 --sims 200 --plies 12]
One JSON line per game: plies/s, simulations/s, ms per round, levels per simulation, valid actions per level, engine errors,
structural validation of the forest afterwards."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from azg_amd import games  # noqa: E402
from azg_amd.selfplay import SelfPlayEngine  # noqa: E402
from hashnet import HashNetHip, HashNetTorch  # noqa: E402


class Args(dict):
    __getattr__ = dict.get


# (name, constructor, share of --games, node capacity in units of numMCTSSims: a Smallworld round spans both players' whole turns, so
#  the round-based clean-up keeps ~30 plies of nodes)
GAMES = [('minivilles', lambda: games.MinivillesGame(2), 1.0, 10), ('abalone', games.AbaloneGame, 1.0, 10), ('thelittleprince', lambda: games.TLPGame(3), 1.0, 10),
         ('botanik', games.BotanikGame, 1.0, 10), ('akropolis', games.AkropolisGame, 0.25, 10), ('smallworld', lambda: games.SmallworldGame(2), 1.0, 80),
         ('smallworld4', lambda: games.SmallworldGame(4), 0.5, 80)]


class MlpNet(torch.nn.Module):
    """A PyTorch module with the reference's forward signature and the size of its small per-game nets (flatten -> 256 -> 256 -> 128 with
    LayerNorm + SiLU, two-layer policy and value heads: the shape of MinivillesNNet V83; the reference ships 2-8 architectures per game,
    they are not restated here).  Random weights: the figure is the cost of a real module per round through nnet.TorchModuleEvaluator."""

    def __init__(self, n_in, A, P):
        super().__init__()
        nn = torch.nn
        self.trunk = nn.Sequential(nn.Linear(n_in, 256), nn.LayerNorm(256), nn.SiLU(), nn.Linear(256, 256), nn.LayerNorm(256), nn.SiLU(),
                                   nn.Linear(256, 128), nn.LayerNorm(128), nn.SiLU())
        self.pi = nn.Sequential(nn.Linear(128, 64), nn.SiLU(), nn.Linear(64, A))
        self.v = nn.Sequential(nn.Linear(128, 64), nn.SiLU(), nn.Linear(64, P))

    def forward(self, x, valid):
        h = self.trunk(x.flatten(1))
        pi = torch.where(valid, self.pi(h), torch.full((), -1e8, device=x.device))
        return torch.log_softmax(pi, dim=1), torch.tanh(self.v(h))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='hash', choices=['hash', 'hashhip', 'mlp', 'engine', 'torchnet'],
                    help='leaf evaluator: integer hash-net as torch ops, the same as one engine kernel (azg_eval_hashnet), MlpNet through TorchModuleEvaluator, '
                         'engine = the game\'s SHIPPED net (minivilles/pretrained_2players.pt V82, thelittleprince/pretrained_3players.pt V83) as one launch of '
                         'the engine\'s MobileNet-1d kernel (nn_mb1d.hip.h); torchnet = the same weights as PyTorch-ROCm ops (nnet.MobileNet1d)')
    ap.add_argument('--md', action='store_true', help='markdown table row instead of JSON')
    ap.add_argument('--games', type=int, default=1024)
    ap.add_argument('--sims', type=int, default=200)
    ap.add_argument('--plies', type=int, default=40, help='timed ply waves (one wave = `sims` lock-step rounds)')
    ap.add_argument('--only', default=None)
    ap.add_argument('--cyc', action='store_true', help='cycle breakdown of k_select per tree and launch (a library built with AZG_DEFINES=AZG_CYC_COUNTERS)')
    a = ap.parse_args()
    for name, make, scale, capf in GAMES:
        if a.only and a.only != name:
            continue
        g = make()
        T = max(64, int(a.games * scale))
        # (Akropolis offers hundreds of placements: with 200 simulations policy-target pruning leaves no count above 1, error bit 64)
        args = Args(numMCTSSims=a.sims, cpuct=1.0, fpu=0.0, universes=1, forced_playouts=name != 'akropolis', prob_fullMCTS=1.0, ratio_fullMCTS=5,
                    dirichletAlpha=0.3, temperature=[1.25, 0.8, 1.0], tempThreshold=6)
        if a.net in ('engine', 'torchnet'):
            tag = {'minivilles': 'minivilles2_v82', 'thelittleprince': 'tlp3_v83'}.get(name)
            if tag is None:
                continue
            from azg_amd import nnet
            base = nnet.MobileNet1d.from_npz(os.path.join(ROOT, 'tests', 'golden', 'weights_%s.npz' % tag), device='cuda:0')
            net = nnet.MobileNet1dHip(base, max_batch=T) if a.net == 'engine' else base
        elif a.net == 'mlp':
            from azg_amd.nnet import TorchModuleEvaluator
            torch.manual_seed(0)
            net = TorchModuleEvaluator(MlpNet(int(g.S), g.A, g.P), g)
        elif a.net == 'hashhip':
            net = HashNetHip(g.P)
            net.bind_outputs(T, g.A, g.P, g.device)          # fixed output buffers: the engine's expansion reads them in place (no copy kernels per round)
        else:
            net = HashNetTorch(g.P)
        eng = SelfPlayEngine(g, net, args, n_games=T, node_capacity=max(2048, capf * a.sims), max_examples=T * 256)
        eng.start()
        eng.run(2 * a.sims)                                    # warm-up: two ply waves (also captures the HIP graph)
        torch.cuda.synchronize()
        s0 = eng.stats()
        t0 = time.perf_counter()
        eng.run(a.plies * a.sims)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = eng.stats()
        sims, plies = s1['sims'] - s0['sims'], s1['plies'] - s0['plies']
        bad = sum(grp.f.validate(verbose=False) for grp in eng.groups)
        if a.md:
            print('| %s | %d | %d | %d | %d | %s | %.0f | %.2f | %.3f | %.2f | %.1f | %d | %d | %.1f |' % (
                name, g.P, g.S, g.A, T, a.net, plies / dt, sims / dt / 1e6, dt / (a.plies * a.sims) * 1e3, (s1['levels'] - s0['levels']) / max(sims, 1),
                (s1['sum_valid_visited'] - s0['sum_valid_visited']) / max(s1['levels'] - s0['levels'], 1), s1['errors'], bad, eng.device_bytes / 1e9), flush=True)
        else:
            print(json.dumps(dict(game=name, players=g.P, state_bytes=g.S, actions=g.A, games=T, sims_per_move=a.sims, plies_per_s=plies / dt,
                                  sims_per_s=sims / dt, ms_per_round=dt / (a.plies * a.sims) * 1e3,
                                  levels_per_sim=(s1['levels'] - s0['levels']) / max(sims, 1),
                                  valid_per_level=(s1['sum_valid_visited'] - s0['sum_valid_visited']) / max(s1['levels'] - s0['levels'], 1),
                                  games_finished=s1['games'], errors=s1['errors'], validate_violations=bad,
                                  forest_gb=eng.device_bytes / 1e9, evaluator=a.net)), flush=True)
        if a.cyc and 'cyc_seg' in s1:
            n = a.plies * a.sims * T
            seg = [s1['cyc_seg'][k] - s0['cyc_seg'][k] for k in range(4)]
            d = {k: s1[k] - s0[k] for k in ('cyc_select', 'cyc_levels', 'cyc_edge', 'cyc_leaf', 'levels', 'sims', 'expansions', 'terminal_hits')}
            print('  %s per tree-launch cycles: select %.0f = levels %.0f (%.2f levels, %.0f each) + edge %.0f (prologue+load_state %.0f make_move %.0f canon+hash %.0f probe %.0f) '
                  '+ leaf %.0f + rest %.0f; expansions %.3f terminal %.3f' % (
                      name, d['cyc_select'] / n, d['cyc_levels'] / n, d['levels'] / n, d['cyc_levels'] / max(1, d['levels']), d['cyc_edge'] / n, seg[0] / n, seg[1] / n, seg[2] / n,
                      seg[3] / n, d['cyc_leaf'] / n, (d['cyc_select'] - d['cyc_levels'] - d['cyc_edge'] - d['cyc_leaf']) / n, d['expansions'] / n, d['terminal_hits'] / n), flush=True)
        for grp in eng.groups:
            grp.f.close()
        del eng
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
