"""Stub of `turtle` (needs tkinter, absent): cogdl/data/sampler.py:3 has a stray `from turtle import pos`."""


def pos(*a, **k):  # pragma: no cover
    raise RuntimeError("turtle stub")
