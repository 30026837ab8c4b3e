#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over csr2csc: the 1 GiB calibration copy, then the hand-written radix
transpose (csrc/radix_transpose.hip) of the Reddit-shaped graph at its true size, five times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.plan import csr2csc  # noqa: E402

dev = "cuda:0"
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
del a, b
g = synth.reddit_like(seed=0, device=dev)
for _ in range(5):
    csr2csc(g.rowptr, g.colind, g.n_cols)
torch.cuda.synchronize()
print("reddit nnz", g.nnz)
