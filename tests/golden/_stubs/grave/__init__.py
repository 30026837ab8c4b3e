"""Stub of `grave` (absent offline; cogdl/pipelines.py:9)."""


def plot_network(*a, **k):  # pragma: no cover
    raise RuntimeError("grave stub")


def use_attributes(*a, **k):  # pragma: no cover
    raise RuntimeError("grave stub")
