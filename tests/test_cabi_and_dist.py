"""CPU-only checks: the C-ABI library loads and exports every symbol include/azg.h declares (no compute without a GPU);
the product fails loudly without a GPU; the multi-GPU example gather works (world_size 2, gloo)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from azg_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'azg.h')).read() + open(os.path.join(ROOT, 'include', 'azg_testaids.h')).read()
    declared = set(re.findall(r'\b(azg_[a-z_0-9]+)\s*\(', hdr))
    product = set(re.findall(r'\b(azg_[a-z_0-9]+)\s*\(', open(os.path.join(ROOT, 'include', 'azg.h')).read()))
    assert not [x for x in product if x.startswith('azg_debug_') or x == 'azg_eval_hashnet']      # test aids live in azg_testaids.h
    assert len(declared) >= 25
    for sym in sorted(declared):
        assert hasattr(L, sym), 'libazg_hip.so does not export %s' % sym
    assert set(_lib.EXPORTS) <= declared
    assert b'gfx950' in L.azg_version()
    S, A, P, rows, cols = _lib.game_info(_lib.SPLENDOR, 2)
    assert (S, A, P, rows, cols) == (392, 81, 2, 56, 7)
    assert _lib.game_info(_lib.SPLENDOR, 4)[0] == 616
    assert _lib.game_info(_lib.SANTORINI, 11)[:3] == (75, 1782, 2)
    assert _lib.game_info(_lib.SANTORINI, 1)[:3] == (75, 162, 2)
    assert _lib.game_info(_lib.AZUL, 0) == (138, 180, 2, 23, 6)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_fails_loudly_without_gpu():
    from azg_amd import games, _lib
    with pytest.raises(_lib.AzgError):
        games.SplendorGame(2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'alpha-zero-general_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'azg_oracle' not in src and 'libazg_oracle' not in src, f


WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from azg_amd.selfplay import gather_examples
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
n = 3 + 2 * rank                                   # ragged: rank 0 has 3 records, rank 1 has 5
boards = torch.full((n, 392), rank + 1, dtype=torch.int8)
pi = torch.arange(n * 81, dtype=torch.float32).reshape(n, 81) + 1000 * rank
z = torch.full((n, 2), float(rank))
out = gather_examples([boards, pi, z])
assert out[0].shape == (8, 392) and out[1].shape == (8, 81) and out[2].shape == (8, 2)
assert (out[0][:3] == 1).all() and (out[0][3:] == 2).all()
assert torch.equal(out[1][3:], torch.arange(5 * 81, dtype=torch.float32).reshape(5, 81) + 1000)
empty = gather_examples([torch.zeros((0 if rank == 0 else 2, 4), dtype=torch.float32)])
assert empty[0].shape == (2, 4)
# gather-to-root (what Coach.learn and bench.py use: Coach.py:150-215 consumes the examples on one rank): ONE grouped send / receive
# of exactly count[k] packed byte rows per rank after the count all_gather, nothing for the other ranks
info = {}
valids = (torch.arange(n * 81).reshape(n, 81) %% 3 == rank).to(torch.uint8)
meta = torch.arange(n * 4, dtype=torch.int32).reshape(n, 4) - 7 * rank
root = gather_examples([boards, pi, z, valids, meta], dst=0, info=info)
assert info['world'] == 2 and info['counts'] == [3, 5] and info['row_bytes'] == 392 + 324 + 8 + 84 + 16
if rank == 0:
    assert [tuple(t.shape) for t in root] == [(8, 392), (8, 81), (8, 2), (8, 81), (8, 4)] and info['bytes_received'] == 8 * info['row_bytes']
    assert (root[0][:3] == 1).all() and (root[0][3:] == 2).all() and root[0].dtype == torch.int8
    assert torch.equal(root[1][3:], torch.arange(5 * 81, dtype=torch.float32).reshape(5, 81) + 1000)
    assert torch.equal(root[3][3:], (torch.arange(5 * 81).reshape(5, 81) %% 3 == 1).to(torch.uint8))
    assert torch.equal(root[4][3:], torch.arange(20, dtype=torch.int32).reshape(5, 4) - 7) and torch.equal(root[4][:3], meta)
else:
    assert all(t.shape[0] == 0 for t in root) and root[1].dtype == torch.float32 and tuple(root[1].shape[1:]) == (81,)
none = gather_examples([torch.zeros((0, 4), dtype=torch.float32)], dst=0)        # nobody has a record: no data exchange at all
assert none[0].shape == (0, 4)
dist.destroy_process_group()
import sys
sys.stdout.write('rank ' + str(rank) + ' ok' + chr(10)); sys.stdout.flush()      # one write: the two ranks share the pipe
'''


def test_example_gather_world2_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(WORKER % ROOT)
    import socket
    with socket.socket() as sk:                      # a free rendezvous port (a fixed one can still be in TIME_WAIT)
        sk.bind(('127.0.0.1', 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', port, str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'rank 0 ok' in r.stdout + r.stderr and 'rank 1 ok' in r.stdout + r.stderr


def test_coach_shares_do_not_depend_on_world_size():
    """Coach's dealing of episodes / arena games over ranks (coach.episode_share / split_range): every global game stream plays the
    same number of episodes whatever the world size (the kernel's per-tree quota, include/azg.h azg_selfplay_start_ex), and the arena
    ranges tile [0, n)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('azg_coach_shares', os.path.join(ROOT, 'alpha-zero-general_amd', 'coach.py'))
    src = open(spec.origin).read()
    ns = {}
    start, end = src.index('def split_range'), src.index('class Coach:')
    exec(src[start:end], ns)                                   # the two pure functions (the module itself imports the engine)
    split_range, episode_share = ns['split_range'], ns['episode_share']

    def tree_quota(q, T, t):                                   # selfplay.hip.h tree_quota
        return q // T + (1 if t < q % T else 0)
    for T in (8, 16, 4096):
        for num_eps in (0, 1, 5, 8, 13, 100, 4096, 10000):
            per_stream = [tree_quota(num_eps, T, t) for t in range(T)]
            for world in (1, 2, 4, 8):
                tl = T // world
                for rank in range(world):
                    q = episode_share(num_eps, T, world, rank)
                    assert q == sum(per_stream[rank * tl:(rank + 1) * tl])
                    assert [tree_quota(q, tl, t) for t in range(tl)] == per_stream[rank * tl:(rank + 1) * tl]
    for n in (0, 1, 6, 30, 31):
        for world in (1, 2, 3, 8):
            rs = [split_range(n, world, r) for r in range(world)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == n
            assert all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(world - 1))


def _amdgpu_kernel_descriptors(path):
    """(kernel name, kernel_code_properties) of every kernel in the gfx950 code objects embedded in the library (plain ELF64 parsing:
    symbols `<kernel>.kd` point at the 64-byte AMDHSA kernel descriptors, kernel_code_properties = u16 at byte 56)"""
    import struct
    blob = open(path, 'rb').read()
    out, i = [], 0
    while True:
        i = blob.find(b'\x7fELF', i)
        if i < 0:
            break
        if struct.unpack_from('<H', blob, i + 18)[0] != 224:        # EM_AMDGPU
            i += 4
            continue
        e = blob[i:]
        shoff, = struct.unpack_from('<Q', e, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from('<HHH', e, 0x3A)
        secs = [struct.unpack_from('<IIQQQQIIQQ', e, shoff + k * shentsize) for k in range(shnum)]
        for s in secs:
            if s[1] != 2:                                               # SHT_SYMTAB
                continue
            stroff = secs[s[6]][4]
            for k in range(s[5] // 24):
                st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from('<IBBHQQ', e, s[4] + 24 * k)
                name = e[stroff + st_name:e.index(b'\0', stroff + st_name)].decode()
                if name.endswith('.kd') and 0 < st_shndx < shnum:
                    sec = secs[st_shndx]
                    kd = sec[4] + (st_value - sec[3])
                    out.append((name[:-3], struct.unpack_from('<H', e, kd + 56)[0]))
        i += 4
    return out


def test_no_kernel_reads_the_dispatch_packet():
    """hipcc's promote-alloca-to-LDS pass makes a kernel read the work-group sizes from the AQL dispatch packet -- a scalar load from the
    queue ring in host memory -- which cost k_select 10-30 k cycles per launch (build.py turns the pass off).  Guard: no
    kernel of the built library has ENABLE_SGPR_DISPATCH_PTR (bit 1 of kernel_code_properties) or the queue pointer (bit 2) set."""
    from azg_amd import _lib
    kds = _amdgpu_kernel_descriptors(_lib.LIB_PATH)
    assert len(kds) > 100, len(kds)
    assert any('k_select' in n for n, _ in kds)
    bad = [n for n, p in kds if p & 0x6]
    assert not bad, bad[:5]
