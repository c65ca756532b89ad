"""Code-generation guard for the pipeline's two persistent kernels (CPU-only: reads the metadata notes of the code objects inside
libazg_hip.so).  Both kernels sit at their register caps (descent: 128 VGPRs x 16 waves; net: 168 x 12) and the allocator is fragile there --
round 6 measured, for changes that touched only COLD code: a post-mortem dump in the time-out branch took the vector spills of the Azul /
Santorini / Splendor descents from 45 / 61 / 17 scratch instructions to 175 / 194 / 75 and Azul at 1600 simulations from 35 to 25.5 k
env-steps/s; thread-derived addresses hoisted out of the net kernel's loop cost the Santorini forward 15 %.  The bounds below are the values of
the measured build plus a little slack: a change that blows them needs a look at the ISA (tools/kernel_resources.sh), not a bigger bound."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
LIB = os.path.join(ROOT, 'alpha-zero-general_amd', 'libazg_hip.so')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def kernel_notes(lib):
    """-> {demangled kernel name: dict(vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds)} over every code object bundled into the library"""
    out = {}
    with tempfile.TemporaryDirectory() as t:
        fat = os.path.join(t, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib, os.devnull])
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, a in enumerate(starts):
            part = os.path.join(t, 'b%d.bin' % k)
            open(part, 'wb').write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            co = os.path.join(t, 'b%d.co' % k)
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--input=' + part,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co, '--unbundle'])
            notes = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], text=True)
            cur = {}
            for ln in notes.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s+(\S+)', ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2)
                if key in ('group_segment_fixed_size', 'private_segment_fixed_size', 'sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'name'):
                    cur[key] = val
                elif key == 'vgpr_spill_count':              # (the last key of a kernel's entry)
                    cur[key] = val
                    if 'name' in cur:
                        out[cur['name']] = dict(vgpr=int(cur.get('vgpr_count', 0)), sgpr=int(cur.get('sgpr_count', 0)), vgpr_spill=int(val),
                                                sgpr_spill=int(cur.get('sgpr_spill_count', 0)), scratch=int(cur.get('private_segment_fixed_size', 0)),
                                                lds=int(cur.get('group_segment_fixed_size', 0)))
                    cur = {}
    names = list(out)
    dem = subprocess.check_output(['c++filt'] + names, text=True).splitlines() if names else []
    return {d: out[n] for n, d in zip(names, dem)}


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, 'llvm-readelf')) and os.path.exists(LIB)), reason='needs the ROCm LLVM tools and the built library')
def test_pipeline_kernels_stay_within_their_register_budgets():
    k = kernel_notes(LIB)
    sel = {n: v for n, v in k.items() if 'k_async_select<' in n}
    net = {n: v for n, v in k.items() if 'k_async_net<' in n}
    assert len(sel) >= 5 and len(net) >= 10, (len(sel), len(net))

    def one(d, frag):
        m = [v for n, v in d.items() if frag in n]
        assert len(m) == 1, (frag, [n for n in d if frag in n])
        return m[0]

    # descent kernels: 16 waves per CU -> at most 128 VGPRs; vector spills (kernel + the out-of-line advance it calls, which spills freely: ~480 B
    # of the scratch size are that cold function's) bounded per game
    # (measured build: 14 / 21 / 27 / 28 / 28 spilled vector registers, 544 - 608 B of scratch)
    for frag, spill_max, scratch_max in (('SplendorDev<2>', 20, 640), ('SplendorDev<3>', 28, 640), ('SplendorDev<4>', 36, 640),
                                         ('SantoriniDev<1>', 36, 672), ('AzulDev', 36, 672)):
        v = one(sel, 'k_async_select<azg::' + frag)
        assert v['vgpr'] <= 128, (frag, v)
        assert v['vgpr_spill'] <= spill_max and v['scratch'] <= scratch_max, (frag, v)
    # net kernels: 12 waves per CU -> at most 168 VGPRs; the forwards of the two north-star games run without scratch memory
    for frag, scratch_max in (('NetV80', 0), ('NetC5', 0)):
        v = one(net, 'k_async_net<azg::' + frag)
        assert v['vgpr'] <= 168 and v['scratch'] <= scratch_max, (frag, v)
    for n, v in net.items():
        assert v['vgpr'] <= 168 and v['scratch'] <= 96, (n, v)
