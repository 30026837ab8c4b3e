#!/usr/bin/env python3
"""A/B of the wave-scope split of medium rows (csrc/rowreduce.h; tuning key 12: -1 = off, 0 = rows of more than 64 edges,
n = rows of more than n edges) on the arxiv-sized graphs: csr_spmm at the widths CogDL's gcn runs, csr_sddmm, the fused
GAT forward.  Kernel time per call (HIP events), % of 8 TB/s by SURVEY 8d's algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
SETTINGS = (-1, 0, 32, 128)
for topo in ("rmat", "uniform"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    n, nnz = g.num_nodes, g.nnz
    deg = g.degrees()
    print("arxiv-%s nnz=%d max_deg=%d; edges in rows of 65..thresh: %.1f %%" % (
        topo, nnz, int(deg.max()), 100.0 * float(deg[(deg > 64) & (deg <= lib.cogdl_hip_long_row_threshold(nnz))].sum()) / nnz), flush=True)
    cases = []
    for f, dt in ((40, torch.float32), (64, torch.float32), (128, torch.float32), (64, torch.bfloat16), (128, torch.bfloat16)):
        x, w = torch.randn(n, f, device=DEV).to(dt), g.weight.to(dt)
        s = x.element_size()
        cases.append(("csr_spmm F=%d %s" % (f, str(dt)[6:]), lambda x=x, w=w: csr_spmm_raw(g.rowptr, g.colind, w, x),
                      nnz * (4 + s + f * s) + n * (4 + f * s)))
    a, b = torch.randn(n, 64, device=DEV), torch.randn(n, 64, device=DEV)
    cases.append(("csr_sddmm F=64", lambda: csr_sddmm_raw(g.rowptr, g.colind, a, b), nnz * (4 + 4 + 2 * 64 * 4 + 4)))
    for dt in (torch.float32, torch.bfloat16):
        ar, ac = torch.randn(n, 8, device=DEV), torch.randn(n, 8, device=DEV)
        ft = torch.randn(n, 8, 8, device=DEV).to(dt)
        s = ft.element_size()
        cases.append(("gat_fwd H=8 F=8 %s" % str(dt)[6:], lambda ft=ft: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, ft),
                      nnz * (4 + 32 + 64 * s) + n * (4 + 64 + 64 * s)))
    for name, fn, nbytes in cases:
        res = []
        for v in SETTINGS:
            lib.cogdl_hip_set_tuning(12, v)
            ms = timeit(fn, 30)
            res.append("%7.1f us (%4.1f %%)" % (ms * 1e3, nbytes / (ms * 1e-3) / 8e12 * 100))
        lib.cogdl_hip_set_tuning(12, 0)
        print("  %-24s off / 64 / 32 / 128:  %s" % (name, "   ".join(res)), flush=True)
