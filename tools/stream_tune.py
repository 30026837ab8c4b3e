#!/usr/bin/env python3
"""GPU experiment (not product): streaming row blocks (tuning key 2 = rows per lane group) for csr_spmm & friends."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
tune = _lib.hip().cogdl_hip_set_tuning
for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    n, nnz = g.num_nodes, g.nnz
    for f in (40, 64, 128, 256):
        x = torch.randn(n, f, device=DEV)
        ref = None
        line = []
        for r in (1, 2, 3, 4, 8, 16, 32):
            tune(2, r)
            ms = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x), 20)
            out = csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
            if ref is None:
                ref = out
            line.append("R=%d %.1f us%s" % (r, ms * 1e3, "" if torch.equal(out, ref) else " MISMATCH"))
        print("csr_spmm %s F=%d: %s" % (topo, f, "  ".join(line)), flush=True)
    x = torch.randn(n, 128, device=DEV)
    y = torch.randn(n, 128, device=DEV)
    ar, ac, feat = torch.randn(n, 8, device=DEV), torch.randn(n, 8, device=DEV), torch.randn(n, 8, 8, device=DEV)
    for name, fn in (("sddmm F=128", lambda: csr_sddmm_raw(g.rowptr, g.colind, y, x)),
                     ("gat_fwd H8F8", lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat)[0])):
        line, ref = [], None
        for r in (1, 2, 4, 8, 15):
            tune(2, r)
            ms = timeit(fn, 20)
            out = fn()
            if ref is None:
                ref = out
            line.append("R=%d %.1f us%s" % (r, ms * 1e3, "" if torch.allclose(out, ref, rtol=1e-5, atol=1e-6) else " MISMATCH"))
        print("%s %s: %s" % (name, topo, "  ".join(line)), flush=True)
tune(2, 1)
