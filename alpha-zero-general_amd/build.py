"""Build the HIP extension in-tree: alpha-zero-general_amd/libazg_hip.so (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libazg_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
         '-I' + os.path.join(HERE, '..', 'include')]
# debugging builds: AZG_DEFINES="AZG_CYC_COUNTERS AZG_NN_PHASE_TIMES" python alpha-zero-general_amd/build.py
FLAGS += ['-D' + d for d in os.environ.get('AZG_DEFINES', '').split()]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + \
        [os.path.join(HERE, '..', 'include', 'azg.h')]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, 'azg.hip'), '-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
