#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: the fused-dropout GAT autograd operator, forward + backward, bf16,
H = 8 x F = 8 and H = 1 x F = 41 on the Reddit-shaped graph (5 steps each) -- which kernels make up the operator's time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func  # noqa: E402

dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
shapes = [(8, 8), (1, 41)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1].split("x"))]
for h, f in shapes:
    ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
    ft = torch.randn(n, h, f, device=dev).bfloat16().requires_grad_()
    grad = torch.randn(n, h, f, device=dev).bfloat16()
    for _ in range(6):
        out = fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, 0.5, seed=3)
        torch.autograd.grad(out, (ar, ac, ft), grad)
    torch.cuda.synchronize()
print("done")
