#!/usr/bin/env python3
"""csr_spmm at the widths CogDL's gcn runs (hidden 64, 40 classes) on the arxiv-shaped graphs: gathers in flight per
lane group (UNROLL 8 / 12 / 16) and lane-group shapes, through cogdl_hip_csr_spmm_variant.  A row of ~15 edges is two
dependent gather batches at UNROLL 8 and one at 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
NAMES = {-1: "default", 6: "VEC2 LPR32 U8", 11: "VEC2 LPR32 U16", 15: "VEC2 LPR32 U12", 2: "VEC4 LPR16 U8", 12: "VEC4 LPR16 U16",
         14: "VEC4 LPR10 U8", 13: "VEC4 LPR10 U16", 3: "VEC2 LPR64 U8", 10: "VEC2 LPR64 U16"}
for f, variants in ((64, (-1, 6, 15, 11, 2, 12)), (40, (-1, 14, 13)), (128, (-1, 3, 10))):
    for topo in ("uniform", "rmat"):
        g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
        x = torch.randn(g.num_nodes, f, device=DEV)
        balg = g.nnz * (8 + f * 4) + g.num_nodes * (4 + f * 4)
        ref = csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
        for rnd in range(2):
            for v in variants:
                out = csr_spmm_raw(g.rowptr, g.colind, g.weight, x, v)
                same = bool(torch.equal(out, ref)) or float((out - ref).abs().max()) < 1e-5 * float(ref.abs().max())
                ms = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, v), 30)
                print("F=%-3d %-7s %-16s %7.1f us  %5.1f %% of 8 TB/s  %s" % (f, topo, NAMES[v], ms * 1e3, balg / ms / 1e6 / 80,
                                                                             "" if same else "MISMATCH"), flush=True)
