"""Build the HIP extension in-tree: alpha-zero-general_amd/libazg_hip.so (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.environ.get('AZG_OUT') or os.path.join(HERE, 'libazg_hip.so')          # AZG_OUT: build a variant for A/B runs
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-pass-failed', '-I' + os.path.join(HERE, '..', 'include')]
# Translation units of the library (compiled concurrently): the forest / env / self-play kernels with their C-ABI, and the net
# kernels with theirs.  (Wave-uniform reads of mutable forest memory are relaxed agent-scope atomic loads -- forest.hip.h
# ld_agent_u32 / load_uniform -- so that the compiler cannot turn them into scalar-cache loads; the blanket alternative,
# -mllvm -amdgpu-scalarize-global-loads=false, was tried in round 2 and miscompiled k_selfplay_advance<AzulDev>.)
# -disable-promote-alloca-to-lds: hipcc's AMDGPUPromoteAlloca pass moves small private arrays with a run-time index (a player's value
# block, ...) into LDS, one slot per work-item, and computes the work-item's linear id from the work-group sizes in the AQL DISPATCH PACKET
# -- a scalar load from the queue ring in HOST memory at the top of k_select's descent: 10-30 k cycles per launch, and the whole of the
# kernel's "slow / fast mode" (47-50 vs 57-62 us; 41.7 us without the load).  With the pass off those 32 bytes are scratch.
UNITS = [('azg.hip', ['-mllvm', '-disable-promote-alloca-to-lds']),
         ('azg_nn.hip', ['-mllvm', '-disable-promote-alloca-to-lds'] + os.environ.get('AZG_NN_FLAGS', '').split()),
         ('azg_async.hip', ['-mllvm', '-disable-promote-alloca-to-lds', '-mllvm', '-disable-machine-licm'] + os.environ.get('AZG_ASYNC_FLAGS', '').split()),
         ('azg_async_sel.hip', ['-mllvm', '-disable-promote-alloca-to-lds'] + os.environ.get('AZG_ASYNC_SEL_FLAGS', '').split())]
# (azg_async.hip / azg_async_sel.hip: the asynchronous tree pipeline, azg_async.hip.h -- the persistent net kernel + the C-ABI / the
# persistent descent kernel.  -disable-machine-licm for the NET kernel only: it is ONE long loop around a body that fills the register
# file; with machine LICM the compiler hoists address constants and zero vectors out of that loop and then spills them -- 34 spilled
# VGPRs / 140 B scratch with it, 4 / 20 B without, the forward 32.3 -> 26.6 us; the descent kernel LOSES with the flag (23.2 -> 30.6 us
# per descent), hence the two units.  AZG_ASYNC_FLAGS / AZG_ASYNC_SEL_FLAGS: further code-generation experiments.)
# (azg_nn.hip also holds the per-CU round kernel, azg_fused.hip.h: 16 tree descents + their net forward in one workgroup)
# debugging builds: AZG_DEFINES="AZG_CYC_COUNTERS AZG_NN_PHASE_TIMES" python alpha-zero-general_amd/build.py
FLAGS += ['-D' + d for d in os.environ.get('AZG_DEFINES', '').split()]
FLAGS += os.environ.get('AZG_EXTRA_FLAGS', '').split()            # e.g. -ftrivial-auto-var-init=pattern when hunting an uninitialised local


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + \
        [os.path.join(HERE, '..', 'include', 'azg.h'), os.path.join(HERE, '..', 'include', 'azg_testaids.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def _unit_stale(obj, dep):
    """an object is up to date when it is newer than every file its compile read (the -MD dependency file next to it)"""
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return True
    t = os.path.getmtime(obj)
    try:
        words = open(dep).read().replace('\\\n', ' ').split()
    except OSError:
        return True
    files = [w for w in words[1:] if not w.endswith(':')]
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=False):
    """force=True (the driver's build check): compile every unit from scratch.  Otherwise units whose sources did not change are
    taken from the object cache (AZG_OBJ_DIR, default build_ab/obj/<flags hash>): a change to the net kernels recompiles azg_nn.hip only."""
    if not force and not needs_build():
        return LIB
    import hashlib
    import shutil
    import tempfile
    cache = None
    if not force:
        cache = os.environ.get('AZG_OBJ_DIR') or os.path.join(HERE, '..', 'build_ab', 'obj')
    tmp = None if cache else tempfile.mkdtemp(prefix='azg_build_')
    objs, procs = [], []
    for src, extra in UNITS:                       # the units compile concurrently
        if cache:                                  # one cache directory per (common flags, this unit's flags): an A/B variant of one
            key = hashlib.sha1(' '.join(FLAGS + [src] + extra).encode()).hexdigest()[:12]     # unit re-uses the other units' objects
            udir = os.path.join(cache, key)
            os.makedirs(udir, exist_ok=True)
        else:
            udir = tmp
        obj = os.path.join(udir, src.replace('.hip', '.o'))
        dep = obj + '.d'
        objs.append(obj)
        if cache and not _unit_stale(obj, dep):
            if verbose:
                print('up to date:', obj)
            continue
        cmd = [HIPCC] + FLAGS + extra + ['-MD', '-MF', dep, '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    if not cache:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--incremental' not in sys.argv, verbose=True))
