"""GPU parity: HIP forest (select / expand_backup / action_probs through the C-ABI) vs the reference's MCTS golden
traces (Numba operand typing) and, node by node, vs the pinned oracle.  N exact; Q compared bit-exactly (tolerance of
the task: 1e-5)."""
import os

import numpy as np
import pytest

from tools_args import MCTS_ARGS

pytestmark = pytest.mark.gpu

F4_VARIANTS = {'abalone': ('abalone', 0), 'akropolis': ('akropolis', 2), 'akropolis3': ('akropolis', 3), 'akropolis4': ('akropolis', 4), 'smallworld': ('smallworld', 2), 'smallworld3': ('smallworld', 3),
               'smallworld4': ('smallworld', 4)}           # SURVEY.md §8 f4 games with the standard (deterministic-step) fixtures
VARIANTS = {'splendor2': ('splendor', 2), 'splendor4': ('splendor', 4), 'santorini1': ('santorini', 1),
            'santorini11': ('santorini', 11), 'azul': ('azul', 0)}


class Args(dict):
    __getattr__ = dict.get


def make(variant):
    from azg_amd import games
    name, v = {**VARIANTS, **F4_VARIANTS}[variant]
    if name == 'azul':
        return games.AzulGame()
    if name == 'abalone':
        return games.AbaloneGame()
    if name == 'akropolis':
        return games.AkropolisGame(v)
    if name == 'smallworld':
        return games.SmallworldGame(v)
    return games.SplendorGame(v) if name == 'splendor' else games.SantoriniGame(v)


@pytest.mark.parametrize('variant,prefix', [(v, 'mcts') for v in list(VARIANTS) + list(F4_VARIANTS)] + [(v, 'mcts800') for v in VARIANTS] + [('azul', 'mcts1600')])
def test_mcts_traces_vs_golden(golden_dir, variant, prefix):
    """`mcts`: 25 / 200 simulations over several argument sets; `mcts800`: the headline search size (800 simulations, the
    checkpoint's args) -- both are outputs of the reference's own MCTS.getActionProb (tools/gen_golden*.py)."""
    import torch
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, '%s_%s_numba.npz' % (prefix, variant)))
    g = make(variant)
    n_cases = len(d['case_sims'])
    # group the cases by identical args so that each group runs as one batched forest
    keys = {}
    for i in range(n_cases):
        k = (int(d['case_sims'][i]), float(d['case_cpuct'][i]), float(d['case_fpu'][i]), int(d['case_universes'][i]),
             int(d['case_forced'][i]))
        keys.setdefault(k, []).append(i)
    for (sims, cpuct, fpu, uni, forced), idxs in keys.items():
        args = Args(numMCTSSims=sims, cpuct=cpuct, fpu=fpu, universes=uni, forced_playouts=bool(forced),
                    prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1])
        T = len(idxs)
        m = BatchedMCTS(g, HashNetTorch(g.P), args, T, node_capacity=sims + 64)
        roots = torch.from_numpy(d['case_root'][idxs]).to(g.device)
        probs, q, full = m.getActionProb(roots, temp=1, force_full_search=True)
        rs = m.forest.root_stats()
        for k, i in enumerate(idxs):
            assert int(rs['Ns'][k]) == int(d['case_Ns'][i]), (variant, i)
            assert np.array_equal(rs['Nsa'][k].cpu().numpy(), d['case_Nsa'][i].astype(np.int32)), (variant, i)
            assert np.allclose(rs['Qsa'][k].cpu().numpy(), d['case_Qsa'][i], rtol=0, atol=1e-5)
            assert np.array_equal(rs['Qsa'][k].cpu().numpy(), d['case_Qsa'][i]), 'Qsa not bit-exact'
            assert float(rs['Qs'][k]) == float(d['case_Qs'][i])
            assert int(rs['n_nodes'][k]) == int(d['case_nodes'][i])
            assert np.array_equal(probs[k].cpu().numpy(), d['case_probs'][i])
            assert np.array_equal(q[k].cpu().numpy(), d['case_q'][i])
            # Ps of valid actions
            va = d['case_Ps'][i] > 0
            assert np.array_equal(rs['Ps'][k].cpu().numpy()[va], d['case_Ps'][i][va])
        m.forest.close()


@pytest.mark.parametrize('variant', ['splendor2', 'santorini11', 'azul', 'abalone', 'akropolis', 'akropolis3', 'akropolis4', 'smallworld', 'smallworld3', 'smallworld4'])
def test_whole_tree_vs_oracle(variant):
    """Every node of the HIP tree equals the oracle's node with the same state key (Ns, Qs, Nsa, Qsa, Ps, Es)."""
    import torch
    import azg_oracle as O
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    from conftest import poison_onchip
    poison_onchip(0xFFFFFFFF)
    g = make(variant)
    name, v = {**VARIANTS, **F4_VARIANTS}[variant]
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL, 'abalone': O.ABALONE, 'akropolis': O.AKROPOLIS, 'smallworld': O.SMALLWORLD}[name], v)
    kw = dict(MCTS_ARGS[variant])
    sims = 400
    T = 8
    roots = np.stack([og.getInitBoard(og.rng(seed=9, stream=i)).reshape(-1) for i in range(T)])
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    m = BatchedMCTS(g, HashNetTorch(g.P), args, T, node_capacity=sims + 64)
    m.getActionProb(torch.from_numpy(roots).to(g.device), temp=1, force_full_search=True)
    for t in range(T):
        om = O.OracleMCTS(og, O.make_args(numMCTSSims=sims, **kw))
        om.getActionProb(roots[t], temp=1, force_full_search=True)
        tree = m.forest.dump_tree(t)
        assert tree['n'] == om.num_nodes()
        for i in range(tree['n']):
            nd = om.node(tree['states'][i])
            assert nd is not None
            assert np.array_equal(nd['Es'], tree['Es'][i])
            assert nd['has_policy'] == bool(tree['has_policy'][i])
            if nd['has_policy']:
                assert nd['Ns'] == int(tree['Ns'][i])
                assert nd['Qs'] == tree['Qs'][i]
                assert np.array_equal(nd['Nsa'].astype(np.int32), tree['Nsa'][i])
                assert np.array_equal(nd['Qsa'], tree['Qsa'][i])
                va = nd['Nsa'] >= 0
                vmask = og.getValidMoves(tree['states'][i], 0)
                assert np.array_equal(nd['Ps'][vmask], tree['Ps'][i][vmask])
    m.forest.close()


@pytest.mark.parametrize('variant', ['splendor2', 'santorini1', 'azul', 'abalone', 'akropolis', 'akropolis4', 'smallworld'])
@pytest.mark.parametrize('small_arena', [False, True])
def test_tree_reuse_sequence_vs_golden(golden_dir, variant, small_arena):
    """Multi-move sequence of the golden set: tree reuse across moves and fast (non-full) searches.  With a small arena
    every search is preceded by the engine's clean-up (the counterpart of MCTS.py:86-91): results must not change."""
    import torch
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, 'mcts_%s_numba.npz' % variant))
    g = make(variant)
    kw = dict(MCTS_ARGS[variant])
    sims = int(d['seq_sims'])
    args = Args(numMCTSSims=sims, prob_fullMCTS=0.0, ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1], **kw)
    n = len(d['seq_action'])
    # round-based clean-up keeps every node that a later transposition could still reach (round >= the root's round), so
    # the arena must hold a few plies' worth of nodes: tight for Splendor (round == ply counter), half the sequence for
    # Santorini / Azul (Azul's round only advances every few plies)
    small_cap = 2 * sims + 72 if variant == 'splendor2' else (sims * n) // 2
    if variant.startswith('smallworld'):
        small_cap = sims * n + 64       # a Smallworld round spans a whole turn of both players: nothing is reclaimable within 12 plies
    m = BatchedMCTS(g, HashNetTorch(g.P), args, 1, node_capacity=small_cap if small_arena else (sims * n + 64))
    for i in range(n):
        root = torch.from_numpy(d['seq_canon'][i:i + 1]).to(g.device)
        full = torch.tensor([1 if i % 3 != 2 else 0], dtype=torch.uint8, device=g.device)
        probs, q, is_full = m.getActionProb(root, temp=1, full=full)
        rs = m.forest.root_stats()
        assert int(is_full[0]) == int(d['seq_full'][i])
        assert int(rs['Ns'][0]) == int(d['seq_Ns'][i]), (variant, i)
        assert np.array_equal(rs['Nsa'][0].cpu().numpy(), d['seq_Nsa'][i].astype(np.int32)), (variant, i)
        assert np.array_equal(probs[0].cpu().numpy(), d['seq_probs'][i]), (variant, i)
    if small_arena and variant == 'splendor2':
        assert m.forest.stats()['gc_runs'] > 0
    m.forest.close()


def test_mcts1600_azul_auto_alpha_noise_vs_golden(golden_dir):
    """BASELINE config 5 (Azul, 1600 simulations, dirichletAlpha = -1: alpha = 10 / n_valid, MCTS.py:188-192): the reference's own
    search with the Dirichlet sample it drew recorded in the fixture (tools/gen_golden_800.py --sims 1600 --noise); the forest gets the
    same sample injected."""
    import torch
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, 'mcts1600_azul_numba.npz'))
    g = make('azul')
    T = len(d['noise_root'])
    args = Args(numMCTSSims=int(d['case_sims'][0]), cpuct=float(d['case_cpuct'][0]), fpu=float(d['case_fpu'][0]),
                universes=int(d['case_universes'][0]), forced_playouts=bool(d['case_forced'][0]), prob_fullMCTS=1.0, ratio_fullMCTS=5,
                dirichletAlpha=-1, temperature=[1.0, 1.0, 1.0])
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=int(d['case_sims'][0]) + 64)
    net = HashNetTorch(g.P)
    nz = torch.from_numpy(d['noise_sample']).to(g.device)
    f.begin_search(torch.from_numpy(d['noise_root']).to(g.device))
    while True:
        f.select(nz, normalised=True)
        if not bool(f.needs_eval.any().item()):
            if f.active() == 0:
                break
            continue
        pi, v = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
        f.expand_backup(pi, v, nz, normalised=True)
    rs = f.root_stats()
    probs, q, _ = f.action_probs(1.0)
    for t in range(T):
        assert int(rs['Ns'][t]) == int(d['noise_Ns'][t])
        assert np.array_equal(rs['Nsa'][t].cpu().numpy(), d['noise_Nsa'][t].astype(np.int32))
        assert np.array_equal(rs['Qsa'][t].cpu().numpy(), d['noise_Qsa'][t])
        va = d['noise_Ps'][t] > 0
        assert np.array_equal(rs['Ps'][t].cpu().numpy()[va], d['noise_Ps'][t][va])
        assert np.array_equal(probs[t].cpu().numpy(), d['noise_probs'][t])
        assert int(rs['n_nodes'][t]) == int(d['noise_nodes'][t])
    f.close()


@pytest.mark.parametrize('temp_root', [1.0, 1.1])
def test_root_dirichlet_noise_injected_vs_oracle(temp_root):
    """applyDirNoise (MCTS.py:187-197) with the SAME Dirichlet sample injected on both sides: new root (noise at
    expansion, :147-149) and, on a second search from the same root, an existing root (:156-160)."""
    import torch
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    og = O.OracleGame(O.SPLENDOR, 2)
    kw = dict(MCTS_ARGS['splendor2'])
    sims, T = 120, 6
    rng = np.random.default_rng(5)
    roots = np.stack([og.getInitBoard(og.rng(seed=21, stream=i)).reshape(-1) for i in range(T)])
    args = Args(numMCTSSims=sims, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0.3,
                temperature=[1.0, 1.0, temp_root], **kw)
    f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=1024)
    net = HashNetTorch(2)
    oracles = [O.OracleMCTS(og, O.make_args(numMCTSSims=sims, dirichletAlpha=0.3, temperature=(1.0, 1.0, temp_root), **kw),
                            dirichlet_noise=True) for _ in range(T)]
    for rep in range(2):
        noise = np.zeros((T, g.A), dtype=np.float64)
        samples = []
        for t in range(T):
            nv = int(og.getValidMoves(roots[t], 0).sum())
            s = rng.dirichlet([0.3] * nv)
            noise[t, :nv] = s
            samples.append(s)
        nz = torch.from_numpy(noise).to(g.device)
        f.begin_search(torch.from_numpy(roots).to(g.device))
        while True:
            f.select(nz, normalised=True)
            if not bool(f.needs_eval.any().item()):
                if f.active() == 0:
                    break
                continue
            pi, v = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
            f.expand_backup(pi, v, nz, normalised=True)
        rs = f.root_stats()
        for t in range(T):
            oracles[t].getActionProb(roots[t], temp=1, force_full_search=True, dir_noise=samples[t])
            nd = oracles[t].node(roots[t])
            va = og.getValidMoves(roots[t], 0)
            ps = rs['Ps'][t].cpu().numpy()
            if temp_root == 1.0:
                assert np.array_equal(ps[va], nd['Ps'][va]), (rep, t)
                assert np.array_equal(rs['Nsa'][t].cpu().numpy(), nd['Nsa'].astype(np.int32)), (rep, t)
                assert np.array_equal(rs['Qsa'][t].cpu().numpy(), nd['Qsa'])
            else:   # pow() in f64 on device vs libm: last-ulp differences allowed
                assert np.allclose(ps[va], nd['Ps'][va], rtol=1e-6, atol=1e-9), (rep, t)
    f.close()


def test_device_dirichlet_sampler_statistics():
    """The engine's own Gamma sampler (no noise tensor): the noised root prior is a proper distribution that differs from
    the clean prior with the moments of 0.75*P + 0.25*Dir(alpha) (mean of the Dirichlet part = 1/n_valid)."""
    import torch
    import azg_oracle as O
    from azg_amd import games
    from azg_amd.forest import Forest
    from hashnet import HashNetTorch
    g = games.SplendorGame(2)
    og = O.OracleGame(O.SPLENDOR, 2)
    kw = dict(MCTS_ARGS['splendor2'])
    T = 512
    root = og.getInitBoard(og.rng(seed=3, stream=0)).reshape(-1)
    roots = torch.from_numpy(np.tile(root, (T, 1))).to(g.device)
    net = HashNetTorch(2)
    res = {}
    for alpha in (0.0, 0.3):
        args = Args(numMCTSSims=4, prob_fullMCTS=1.0, dirichletAlpha=alpha, temperature=[1.0, 1.0, 1.0], **kw)
        f = Forest(g.GAME_ID, g.variant, T, args, node_capacity=64, rng_seed=11)
        f.begin_search(roots)
        f.select(device_noise=True)
        pi, v = net.predict_batch(f.leaf_states.view((T,) + f.board_shape()), f.leaf_valid.bool())
        f.expand_backup(pi, v, device_noise=True)
        f.select(device_noise=True)        # k_root_noise runs at the head of the next select
        res[alpha] = f.root_stats()['Ps'].cpu().numpy().astype(np.float64)
        f.close()
    clean, noisy = res[0.0], res[0.3]
    va = og.getValidMoves(root, 0)
    nv = int(va.sum())
    assert np.allclose(noisy.sum(axis=1), 1.0, atol=1e-5) and np.all(noisy[:, ~va] == 0)
    d = (noisy[:, va] - 0.75 * clean[:, va]) / 0.25           # the Dirichlet part per tree
    assert np.all(d > -1e-6) and np.allclose(d.sum(axis=1), 1.0, atol=1e-4)
    assert abs(d.mean() - 1.0 / nv) < 1e-3
    # Var of a Dirichlet(alpha) marginal: (1/n)(1-1/n)/(n*alpha+1)
    expect_var = (1.0 / nv) * (1 - 1.0 / nv) / (nv * 0.3 + 1)
    assert abs(d.var() - expect_var) / expect_var < 0.15
    assert len({tuple(np.round(x, 6)) for x in d[:32]}) == 32      # different trees, different samples


@pytest.mark.parametrize('budget', [dict(level_budget=1), dict(level_budget=3), dict(work_budget=2), dict(work_budget=7)])
def test_level_budget_is_pure_scheduling(golden_dir, budget):
    """A per-launch level / work budget (descents parked and resumed across launches) must not change any statistic."""
    import torch
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    d = np.load(os.path.join(golden_dir, 'mcts_splendor2_numba.npz'))
    g = make('splendor2')
    idxs = [i for i in range(len(d['case_sims'])) if int(d['case_sims'][i]) == 200 and int(d['case_universes'][i]) == 3
            and int(d['case_forced'][i]) == 1 and abs(float(d['case_fpu'][i]) - 0.0593) < 1e-9]
    assert len(idxs) >= 2
    args = Args(numMCTSSims=200, cpuct=0.8, fpu=0.0593, universes=3, forced_playouts=True, prob_fullMCTS=1.0,
                ratio_fullMCTS=5, dirichletAlpha=0, temperature=[1, 1, 1])
    m = BatchedMCTS(g, HashNetTorch(g.P), args, len(idxs), node_capacity=512, **budget)
    probs, q, _ = m.getActionProb(torch.from_numpy(d['case_root'][idxs]).to(g.device), temp=1, force_full_search=True)
    rs = m.forest.root_stats()
    for k, i in enumerate(idxs):
        assert np.array_equal(rs['Nsa'][k].cpu().numpy(), d['case_Nsa'][i].astype(np.int32))
        assert np.array_equal(rs['Qsa'][k].cpu().numpy(), d['case_Qsa'][i])
        assert np.array_equal(probs[k].cpu().numpy(), d['case_probs'][i])
        assert int(rs['n_nodes'][k]) == int(d['case_nodes'][i])
    assert m.forest.validate() == 0
    m.forest.close()


# RNG-free known answers of the reference's MCTS with the integer hash-net from the deterministic
# start states of Appendix C.1 (no fixture file involved) -- the HIP forest's twin of tests/test_oracle_known_answers.py
C3 = [
    ('splendor2', (0, 0), dict(cpuct=0.8, fpu=0.0593, universes=3), 25, 25, 24, 0.2179533839225769, '831b8a698a50baae', 1,
     [(13, 22, 0.3160876), (30, 1, -0.7252549), (16, 1, -0.0159627)]),
    ('splendor2', (0, 0), dict(cpuct=0.8, fpu=0.0593, universes=3), 800, 800, 799, -0.0020540114492177963, '617f8d15987a642e', 18,
     [(13, 217, 0.0368961), (25, 179, 0.0762485), (46, 88, 0.0316085), (53, 85, 0.030627), (16, 65, 0.0238493)]),
    ('santorini1', (0, 0), dict(cpuct=1.1, fpu=0.03, universes=0), 25, 25, 24, -0.17132920026779175, '1b51c44b02553a62', 1,
     [(156, 19, -0.0699613)]),
    ('santorini1', (0, 0), dict(cpuct=1.1, fpu=0.03, universes=0), 800, 800, 799, -0.018680008128285408, 'f6a86e2b69d5f887', 31,
     [(69, 198, 0.0326832), (127, 61, 0.021105), (152, 60, 0.0304925), (100, 42, 0.0265984)]),
    ('santorini11', (1, 5), dict(cpuct=1.1, fpu=0.03, universes=0), 800, 800, 799, 0.0020819352939724922, '5bb85ebbe5d154c8', 37,
     [(101, 104, 0.0759137), (1, 102, 0.071334), (10, 45, 0.1134226), (28, 41, 0.0644857)]),
    ('azul', (0, 0), dict(cpuct=0.5, fpu=0.05, universes=1), 800, 800, 799, -0.01931973174214363, 'f0e32ed35ac9f702', 32,
     [(122, 169, 0.0459814), (34, 64, -0.0727988), (169, 53, 0.0370114), (41, 52, 0.0458253)]),
]


@pytest.mark.parametrize('case', C3, ids=lambda c: '%s-%d' % (c[0], c[3]))
def test_c3_known_answers_on_the_forest(case):
    import hashlib
    import torch
    import azg_oracle as O
    from azg_amd.mcts import BatchedMCTS
    from hashnet import HashNetTorch
    variant, gods, kw, sims, nodes, Ns, Qs, nsa_sha, nonzero, top = case
    name, v = VARIANTS[variant]
    og = O.OracleGame({'splendor': O.SPLENDOR, 'santorini': O.SANTORINI, 'azul': O.AZUL}[name], v)
    root = og.known_start(*gods).reshape(1, -1)           # Appendix C.1 start state (the oracle only builds the INPUT here)
    g = make(variant)
    args = Args(numMCTSSims=sims, forced_playouts=True, prob_fullMCTS=1.0, ratio_fullMCTS=5, dirichletAlpha=0,
                temperature=[1, 1, 1], **kw)
    m = BatchedMCTS(g, HashNetTorch(g.P), args, 1, node_capacity=sims + 64)
    probs, q, _ = m.getActionProb(torch.from_numpy(root).to(g.device), temp=1, force_full_search=True)
    rs = m.forest.root_stats()
    nsa = rs['Nsa'][0].cpu().numpy().astype(np.int64)
    qsa = rs['Qsa'][0].cpu().numpy()
    assert int(rs['n_nodes'][0]) == nodes and int(rs['Ns'][0]) == Ns
    assert float(rs['Qs'][0]) == float(np.float32(Qs))
    assert hashlib.sha256(nsa.tobytes()).hexdigest()[:16] == nsa_sha
    for a, n, qv in top:
        assert nsa[a] == n and abs(qsa[a] - qv) < 1e-6, (a, nsa[a], qsa[a])
    assert int((probs[0] > 0).sum()) == nonzero
    qq = q[0].cpu().numpy()
    assert qq[0] == np.float32(Qs) and qq[1] == -np.float32(Qs)
    m.forest.close()


@pytest.mark.gpu
def test_device_hashnet_equals_torch_hashnet():
    """azg_eval_hashnet (the stand-in evaluator as one engine kernel) == tests/hashnet.py HashNetTorch bit for bit, on random boards of
    three plugins incl. negative board bytes and a 3-player value vector"""
    import torch
    from hashnet import HashNetHip, HashNetTorch
    gen = torch.Generator().manual_seed(5)
    for S, A, P, T in ((320, 131, 2, 257), (825, 9, 3, 64), (2310, 428, 2, 33)):
        boards = torch.randint(-128, 128, (T, S), dtype=torch.int8, generator=gen).cuda()
        valids = (torch.rand((T, A), generator=gen) < 0.4).to(torch.uint8)
        valids[:, 0] = 1
        valids = valids.cuda()
        pi0, v0 = HashNetTorch(P).predict_batch(boards, valids)
        pi1, v1 = HashNetHip(P).predict_batch(boards, valids)
        torch.cuda.synchronize()
        assert torch.equal(pi0, pi1) and torch.equal(v0, v1)
