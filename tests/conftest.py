import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

GOLDEN = os.environ.get('AZG_GOLDEN_DIR', os.path.join(ROOT, 'tests', 'golden'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN



def poison_onchip(pattern=0xFFFFFFFF):
    """debug hook of the engine: leave `pattern` in every CU's LDS and in the default queue's scratch memory, so that a kernel reading
    on-chip memory it never wrote fails its parity test instead of depending on what ran before"""
    import ctypes
    from azg_amd._lib import lib
    lib().azg_debug_poison_onchip.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
    lib().azg_debug_poison_onchip(pattern, None)
