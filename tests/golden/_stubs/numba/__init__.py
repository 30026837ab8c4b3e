"""Stub of `numba` (absent offline): njit/prange become no-ops (cogdl/utils/sampling.py:1)."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


jit = njit
prange = range


class _Type:
    """numba.float32 & co.: only ever used as annotations inside njit(locals=...) (cogdl/utils/ppr_utils.py:7)."""

    def __init__(self, name):
        self.name = name

    def __call__(self, v):
        return v

    def __getitem__(self, item):
        return self


float32, float64, int32, int64, boolean = (_Type(n) for n in ("float32", "float64", "int32", "int64", "boolean"))
