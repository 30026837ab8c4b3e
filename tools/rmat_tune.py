#!/usr/bin/env python3
"""csr_spmm on the power-law arxiv-shaped graph: long-row threshold (tuning key 1) and long-row grid (key 3) sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
g = synth.arxiv_like(seed=0, topology="rmat").to(DEV)
deg = g.degrees()
print("arxiv-rmat nnz=%d max_deg=%d; edges in rows > 128: %.1f%%, > 512: %.1f%%, > 2048: %.1f%%" % (
    g.nnz, int(deg.max()), *[100.0 * float(deg[deg > t].sum()) / g.nnz for t in (128, 512, 2048)]), flush=True)
lib = _lib.hip()
for f, dt in ((64, torch.float32), (64, torch.bfloat16), (128, torch.float32)):
    x = torch.randn(g.num_nodes, f, device=DEV).to(dt)
    w = g.weight.to(dt)
    for thr in (0, 64, 256, 512, 1024, 4096, 1 << 20):
        res = []
        for grid in (512, 1024, 2040):
            lib.cogdl_hip_set_tuning(1, thr)
            lib.cogdl_hip_set_tuning(3, grid)
            res.append("%7.1f" % (timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 20) * 1e3))
        print("F=%-3d %-8s thresh %-7s  us at long grid 512/1024/2040: %s" % (f, str(dt)[6:], thr or "auto", " ".join(res)), flush=True)
lib.cogdl_hip_set_tuning(1, 0)
lib.cogdl_hip_set_tuning(3, 1024)
