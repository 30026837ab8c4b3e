#!/usr/bin/env python3
"""Where does the flat edge_softmax kernel's time go?  Timing-only variants (tuning key 9, wrong results for bits 1-3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
cases = [("arxiv-uniform", synth.arxiv_like(seed=0).to(DEV)), ("arxiv-rmat", synth.arxiv_like(seed=0, topology="rmat").to(DEV)),
         ("reddit", synth.reddit_like(seed=0, device=DEV))]
for name, g in cases:
    for h in (1, 8):
        a = torch.randn(g.nnz, h, device=DEV)
        sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
        for dbg, label in ((0, "default"), (1, "blockIdx tiles"), (3, "+no exchange"), (7, "+no search"), (15, "+no processing (copy)")):
            lib.cogdl_hip_set_tuning(9, dbg)
            f = timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a), 10)
            b = timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, a), 10)
            print("%-14s H=%d %-24s fwd %8.1f us (%5.0f GB/s)  bwd %8.1f us (%5.0f GB/s)" % (
                name, h, label, f * 1e3, g.nnz * h * 8 / f / 1e6, b * 1e3, g.nnz * h * 12 / b / 1e6), flush=True)
        lib.cogdl_hip_set_tuning(9, 0)
