"""Stub `numba` used ONLY in the build container to import the reference in pure-Python mode
(the reference's own documented NUMBA_DISABLE_JIT=1 debug mode, README.md:169-171).
Never shipped to the GPU box; nothing under tests -m gpu / bench.py / smoke() imports it."""


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit = _identity_decorator


class _T:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


int8 = uint8 = int16 = uint16 = int32 = uint32 = int64 = uint64 = float32 = float64 = boolean = b1 = _T()

from . import experimental  # noqa: E402,F401
