"""Stub of `numba` (absent offline): njit/prange become no-ops (cogdl/utils/sampling.py:1)."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
prange = range
