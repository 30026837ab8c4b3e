#!/usr/bin/env python3
"""Round 6: long-row grid (tuning key 3) x threshold (key 1) on the Reddit-shaped graph after the VALU diet of the row-reduce
kernels (fewer registers -> more resident workgroups: does a larger long-row grid pay now?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

dev = "cuda:0"
lib = _lib.hip()
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
x, w = torch.randn(n, 64, device=dev).bfloat16(), g.weight.bfloat16()
ar, ac = torch.randn(n, 8, device=dev), torch.randn(n, 8, device=dev)
feat = torch.randn(n, 8, 8, device=dev).bfloat16()
cases = [("csr_spmm bf16 F=64", lambda: csr_spmm_raw(g.rowptr, g.colind, w, x)),
         ("gat_fwd bf16 H8F8", lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat))]
for name, fn in cases:
    for thr in (512, 1024):
        out = []
        for grid in (1024, 1280, 1536, 2040):
            lib.cogdl_hip_set_tuning(1, thr)
            lib.cogdl_hip_set_tuning(3, grid)
            out.append("%8.1f" % (timeit(fn, 8) * 1e3))
        print("%-20s thresh %-5d us at long grid 1024/1280/1536/2040: %s" % (name, thr, " ".join(out)), flush=True)
    lib.cogdl_hip_set_tuning(1, 0)
    lib.cogdl_hip_set_tuning(3, 1024)
