#!/usr/bin/env python3
"""Round 6: the long-row threshold (tuning key 1) and the long-row grid (key 3) were tuned on the arxiv-sized R-MAT graph
only (profiles/r04_rmat_tune.txt).  The Reddit-shaped graph has 82 % of its edges in rows above the automatic threshold
(1024), i.e. in the <= 1024 long-row workgroups = at most HALF the chip's wave slots.  Sweep both on that graph: csr_spmm
F = 64 bf16 / fp32, fused GAT forward bf16 H = 8 x F = 8 and H = 1 x F = 48; plus who takes the time (key 13)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
g = synth.reddit_like(seed=0, device=DEV, norm="sym")
n = g.num_nodes
deg = g.degrees()
print("reddit-shaped nnz=%d max_deg=%d; edges in rows > 128: %.1f%%, > 512: %.1f%%, > 1024: %.1f%%, > 4096: %.1f%%" % (
    g.nnz, int(deg.max()), *[100.0 * float(deg[deg > t].sum()) / g.nnz for t in (128, 512, 1024, 4096)]), flush=True)
cases = []
for f, dt in ((64, torch.bfloat16), (64, torch.float32)):
    x, w = torch.randn(n, f, device=DEV).to(dt), g.weight.to(dt)
    cases.append(("csr_spmm F=%d %s" % (f, str(dt)[6:]), lambda x=x, w=w: csr_spmm_raw(g.rowptr, g.colind, w, x)))
for h, f in ((8, 8), (1, 48)):
    ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
    feat = torch.randn(n, h, f, device=DEV).to(torch.bfloat16)
    cases.append(("gat_fwd H=%d F=%d bf16" % (h, f), lambda ar=ar, ac=ac, feat=feat: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat)))
for name, fn in cases:
    res = []
    for dbg in (0, 1, 2):
        lib.cogdl_hip_set_tuning(13, dbg)
        res.append(timeit(fn, 8) * 1e3)
    lib.cogdl_hip_set_tuning(13, 0)
    print("%-24s default %8.1f us   long-row workgroups alone %8.1f   row blocks alone %8.1f" % (name, *res), flush=True)
    for thr in (128, 256, 512, 1024, 2048, 4096):
        out = []
        for grid in (512, 1024, 2040):
            lib.cogdl_hip_set_tuning(1, thr)
            lib.cogdl_hip_set_tuning(3, grid)
            out.append("%8.1f" % (timeit(fn, 8) * 1e3))
        print("   thresh %-5d  us at long grid 512/1024/2040: %s" % (thr, " ".join(out)), flush=True)
    lib.cogdl_hip_set_tuning(1, 0)
    lib.cogdl_hip_set_tuning(3, 1024)
