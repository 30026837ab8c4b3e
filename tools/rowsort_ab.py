#!/usr/bin/env python3
"""A/B of the degree-sorted row assignment inside a workgroup (tuning key 2: 1 = off) on the arxiv-shaped graphs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from cogdl_amd.operators.mhspmm import mhspmm_raw  # noqa: E402
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    n = g.num_nodes
    cases = []
    for f, dt in ((16, torch.float32), (40, torch.float32), (64, torch.float32), (64, torch.bfloat16), (128, torch.bfloat16), (128, torch.float32)):
        x = torch.randn(n, f, device=DEV).to(dt)
        w = g.weight.to(dt)
        cases.append(("csr_spmm F=%d %s" % (f, str(dt)[6:]), lambda x=x, w=w: csr_spmm_raw(g.rowptr, g.colind, w, x)))
    y64, x64 = torch.randn(n, 64, device=DEV), torch.randn(n, 64, device=DEV)
    cases.append(("csr_sddmm F=64", lambda: csr_sddmm_raw(g.rowptr, g.colind, y64, x64)))
    att = torch.rand(g.nnz, 8, device=DEV)
    feat = torch.randn(n, 8, 8, device=DEV)
    cases.append(("mhspmm H=8 F=8", lambda: mhspmm_raw(g.rowptr, g.colind, att, feat)))
    ar, ac = torch.randn(n, 8, device=DEV), torch.randn(n, 8, device=DEV)
    cases.append(("gat_fwd H=8 F=8", lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat)))
    for name, fn in cases:
        res = []
        for off in (1, 0):
            lib.cogdl_hip_set_tuning(2, off)
            res.append(timeit(fn, 20) * 1e3)
        print("arxiv-%-8s %-24s natural %7.1f us   degree-sorted %7.1f us   (%+.1f%%)" % (
            topo, name, res[0], res[1], 100 * (res[1] / res[0] - 1)), flush=True)
lib.cogdl_hip_set_tuning(2, 0)
