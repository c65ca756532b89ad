"""RNG-free known answers captured from the reference: start-state hashes (C.1), env trajectory
digests (C.2) and hash-net MCTS root statistics (C.3).  Independent of the .npz fixtures."""
import hashlib

import numpy as np
import pytest

import azg_oracle as O

MAGIC = [31416, 1, 14142, 42, 27183, 2, 16180, 7]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.int8).tobytes()).hexdigest()[:16]


def trajectory(g, start):
    st, p, t, nv, acts = start.copy(), 0, 0, 0, []
    dig = hashlib.sha256()
    while True:
        e = g.getGameEnded(st, p)
        if e.any():
            break
        v = g.getValidMoves(st, p)
        idx = np.flatnonzero(v)
        nv += len(idx)
        a = int(idx[(7 * t + 3) % len(idx)])
        st, p2 = g.getNextState(st, p, a, random_seed=MAGIC[t % 8])
        dig.update(np.packbits(v.astype(np.uint8)).tobytes() + a.to_bytes(2, 'little') + int(p2).to_bytes(1, 'little') +
                   st.tobytes())
        acts.append(a)
        p = p2
        t += 1
    return t, nv, acts[:8], sha(st), dig.hexdigest()[:16], list(e)


C2 = [
    ((O.SPLENDOR, 2), (0, 0), '64b3d7666594057a', 124, 3506, [15, 22, 32, 39, 46, 52, 65, 12], '490728ba2d75a26b', '974825d0862a312a', [1, -1]),
    ((O.SPLENDOR, 4), (0, 0), '519651973bf647db', 248, 7018, [15, 22, 32, 39, 46, 53, 62, 15], 'bd8e94aa669924b1', 'dcb441bc5295fcb8', [-1, -1, 1, -1]),
    ((O.SANTORINI, 1), (0, 0), '9ab10b96db1277d7', 68, 2240, [7, 23, 65, 75, 80, 134, 84, 136], '8e16658478d813f1', '2f765fcc0c900383', [-1, 1]),
    ((O.SANTORINI, 11), (1, 5), 'd77318997cf73558', 69, 2923, [7, 23, 65, 75, 80, 939, 894, 936], 'aa76a029590e9eae', 'abba95a3c54706ce', [1, -1]),
    ((O.SANTORINI, 11), (2, 6), None, 53, 1600, [7, 509, 42, 75, 1404, 932, 894, 521], 'cc5ce8b4ff1a6fd4', 'a509e4e808b04ea3', [-1, 1]),
    ((O.SANTORINI, 11), (7, 10), None, 54, 2669, [7, 23, 50, 849, 75, 63, 947, 73], '05a8796c6ae75459', 'd553c04776663f4d', [-1, 1]),
    ((O.SANTORINI, 11), (3, 9), None, 46, 2106, [7, 23, 65, 75, 80, 944, 250, 3], '174d42d233374269', 'c2e5b7f81bd4c1f7', [-1, 1]),
    ((O.SANTORINI, 11), (8, 4), None, 23, 2035, [7, 23, 65, 75, 80, 891, 907, 916], '32d0d9278fa27e47', '40fbfce53c2f7f3e', [1, -1]),
    ((O.AZUL, 0), (0, 0), 'd1e0bbcd4c4277f0', 63, 1435, [33, 16, 73, 106, 150, 140, 20, 1], 'ac77a8d87f636d12', '62a229650d22c744', [1, -1]),
]


@pytest.mark.parametrize('case', C2)
def test_c1_c2_env_known_answers(case):
    gv, gods, start_sha, plies, nvalid, acts, final_sha, digest, result = case
    g = O.OracleGame(*gv)
    start = g.known_start(*gods)
    if start_sha:
        assert sha(start) == start_sha
    t, nv, a8, fs, dg, res = trajectory(g, start)
    assert (t, nv, a8, fs, dg) == (plies, nvalid, acts, final_sha, digest)
    assert [float(x) for x in res] == [float(x) for x in result]


C3 = [
    ((O.SPLENDOR, 2), (0, 0), dict(cpuct=0.8, fpu=0.0593, universes=3), 25, 25, 24, 0.2179533839225769, '831b8a698a50baae', 1),
    ((O.SPLENDOR, 2), (0, 0), dict(cpuct=0.8, fpu=0.0593, universes=3), 800, 800, 799, -0.0020540114492177963, '617f8d15987a642e', 18),
    ((O.SANTORINI, 1), (0, 0), dict(cpuct=1.1, fpu=0.03, universes=0), 25, 25, 24, -0.17132920026779175, '1b51c44b02553a62', 1),
    ((O.SANTORINI, 1), (0, 0), dict(cpuct=1.1, fpu=0.03, universes=0), 800, 800, 799, -0.018680008128285408, 'f6a86e2b69d5f887', 31),
    ((O.SANTORINI, 11), (1, 5), dict(cpuct=1.1, fpu=0.03, universes=0), 800, 800, 799, 0.0020819352939724922, '5bb85ebbe5d154c8', 37),
    ((O.AZUL, 0), (0, 0), dict(cpuct=0.5, fpu=0.05, universes=1), 800, 800, 799, -0.01931973174214363, 'f0e32ed35ac9f702', 32),
]


@pytest.mark.parametrize('typing', [False, True])
@pytest.mark.parametrize('case', C3)
def test_c3_mcts_known_answers(case, typing):
    gv, gods, kw, sims, nodes, Ns, Qs, nsa_sha, nonzero = case
    g = O.OracleGame(*gv)
    root = g.known_start(*gods)
    m = O.OracleMCTS(g, O.make_args(numMCTSSims=sims, forced_playouts=True, numpy2_scalar_typing=typing, **kw))
    probs, q, _ = m.getActionProb(root, temp=1, force_full_search=True)
    nd = m.node(root)
    assert m.num_nodes() == nodes and nd['Ns'] == Ns
    assert float(nd['Qs']) == float(np.float32(Qs))
    assert hashlib.sha256(nd['Nsa'].astype(np.int64).tobytes()).hexdigest()[:16] == nsa_sha
    assert int((probs > 0).sum()) == nonzero
    assert list(q) == [nd['Qs'], -nd['Qs']]
