#!/usr/bin/env python3
"""Workload for `rocprofv3 --pmc <memory-pipeline counters>` passes over the kernels BASELINE configs[2] actually runs (the
XCD-partitioned plan path of the fused GAT operator and of the 16-bit csr_spmm on the Reddit-shaped graph): which unit is
busy while the launch lasts -- the texture addresser (TA: one wave64 memory instruction per 16 cycles and CU), the L1 (TCP),
the L2 (TCC) or the vector ALUs?  tools/sq_summarize.py folds the CSV (any counter names)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
import cogdl_amd.operators.fused_gat as fg  # noqa: E402
from cogdl_amd.operators.spmm import csrspmm  # noqa: E402

dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
x = torch.randn(n, 64, device=dev).bfloat16()
w = g.weight.bfloat16()
for _ in range(3):
    csrspmm(g.rowptr, g.colind, x, w)
torch.cuda.synchronize()
for h, f, p in ((8, 8, 0.0), (8, 8, 0.5), (1, 41, 0.5)):
    ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
    ft = torch.randn(n, h, f, device=dev).bfloat16().requires_grad_()
    grad = torch.randn(n, h, f, device=dev).bfloat16()
    for _ in range(3):
        o = fg.fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, p, seed=3)
        torch.autograd.grad(o, (ar, ac, ft), grad)
    torch.cuda.synchronize()
print("done")
