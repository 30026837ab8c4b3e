#!/usr/bin/env python3
"""Quick timing of the configs[2] kernels on the Reddit-shaped graph (bf16): fused GAT forward / backward with and without
dropout at H = 8 x F = 8 and H = 1 x F = 41, csr_spmm F = 64, mhspmm.  Correctness is the test suite's job; this prints
microseconds per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func, gat_forward  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
for dt in (torch.bfloat16, torch.float32):
    x, w = torch.randn(n, 64, device=dev).to(dt), g.weight.to(dt)
    print("csr_spmm F=64 %-8s %8.1f us" % (str(dt)[6:], timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10) * 1e3), flush=True)
for h, f in ((8, 8), (1, 41)):
    ar, ac = torch.randn(n, h, device=dev), torch.randn(n, h, device=dev)
    for dt in (torch.bfloat16, torch.float32):
        feat = torch.randn(n, h, f, device=dev).to(dt)
        grad = torch.randn(n, h, f, device=dev).to(dt)
        print("gat_fwd H=%d F=%d %-8s %8.1f us" % (h, f, str(dt)[6:], timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), 10) * 1e3), flush=True)
        for p in (0.0, 0.5):
            a, c, ft = ar.clone().requires_grad_(), ac.clone().requires_grad_(), feat.clone().requires_grad_()
            t_f = timeit(lambda: fused_gat_dropout_func(a, c, g.rowptr, g.colind, 0.2, ft, p, seed=3), 10)
            out = fused_gat_dropout_func(a, c, g.rowptr, g.colind, 0.2, ft, p, seed=3)
            t_b = timeit(lambda: torch.autograd.grad(out, (a, c, ft), grad, retain_graph=True), 10)
            print("   autograd op p=%.1f   forward %8.1f us   backward %8.1f us" % (p, t_f * 1e3, t_b * 1e3), flush=True)
