#!/usr/bin/env python3
"""Per-row cost of the row-reduce engine: csr_spmm on graphs of 169,343 rows with EXACTLY d edges per row (random
columns), d = 0 .. 32: if the time barely moves with d, a launch is bound by the dependent chain of a row
(rowptr -> colind/val -> gather -> store), not by bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
n = 169_343
for f, dt in ((64, torch.float32), (128, torch.float32), (64, torch.bfloat16)):
    x = torch.randn(n, f, device=DEV).to(dt)
    out = []
    for d in (0, 1, 2, 4, 8, 15, 16, 17, 32):
        rowptr = (torch.arange(n + 1, device=DEV) * d).int()
        colind = torch.randint(0, n, (n * d,), device=DEV).int()
        w = torch.rand(n * d, device=DEV).to(dt)
        us = timeit(lambda: csr_spmm_raw(rowptr, colind, w, x), 30) * 1e3
        out.append("d=%-2d %6.1f us" % (d, us))
    print("F=%-3d %-8s %s" % (f, str(dt)[6:], "  ".join(out)), flush=True)
