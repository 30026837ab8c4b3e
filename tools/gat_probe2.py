#!/usr/bin/env python3
"""Fused GAT on the Reddit-shaped graph, configs[2]'s two layer shapes as the autograd operator runs them (rows padded
to 16 bytes: H=8 x F=8 and H=1 x F=48 in bf16): forward kernel choice (tuning key 5: 1 = edge-wise online softmax,
2 = chunk-wise) and vector cap (key 4), forward and backward, with and without dropout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func, gat_forward  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
g = synth.reddit_like(seed=0, device=DEV)
n, nnz = g.num_nodes, g.nnz
for h, f, dt in ((1, 48, torch.bfloat16), (1, 44, torch.float32), (8, 8, torch.bfloat16), (8, 8, torch.float32), (1, 64, torch.bfloat16)):
    s = 2 if dt == torch.bfloat16 else 4
    ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
    feat = torch.randn(n, h, f, device=DEV).to(dt)
    b_fwd = nnz * (4 + 4 * h + h * f * s) + n * (4 + 8 * h + h * f * s)
    for p in (0.0, 0.5):
        res = []
        for kern in (1, 2):
            for vcap in (0, 4):
                lib.cogdl_hip_set_tuning(5, kern)
                lib.cogdl_hip_set_tuning(4, vcap)
                ms = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat, p, 1), 5)
                res.append("%s/vec%s %6.2f ms (%4.1f %%)" % ("online" if kern == 1 else "chunk", vcap or "max", ms, b_fwd / (ms * 1e-3) / 8e12 * 100))
        lib.cogdl_hip_set_tuning(5, 0)
        lib.cogdl_hip_set_tuning(4, 0)
        print("fwd H=%d F=%d %s p=%.1f:  %s" % (h, f, str(dt)[6:], p, "   ".join(res)), flush=True)

# backward: vector cap (tuning key 4) -- fat lanes (fewer repeats of the per-edge attention maths) against more, thinner lanes
# (lower register count per lane, more waves per SIMD)
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func  # noqa: E402
for h, f, dt in ((8, 8, torch.bfloat16), (1, 48, torch.bfloat16), (8, 8, torch.float32)):
    ar, ac = torch.randn(n, h, device=DEV).requires_grad_(), torch.randn(n, h, device=DEV).requires_grad_()
    feat = torch.randn(n, h, f, device=DEV).to(dt).requires_grad_()
    gout = torch.randn(n, h, f, device=DEV).to(dt)
    for p in (0.0, 0.5):
        res = []
        for vcap in (0, 4, 2):
            lib.cogdl_hip_set_tuning(4, vcap)
            o = fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, feat, p, seed=1)
            ms = timeit(lambda: torch.autograd.grad(o, (ar, ac, feat), gout, retain_graph=True), 5)
            res.append("vec%s %6.2f ms" % (vcap or "max", ms))
            del o
        lib.cogdl_hip_set_tuning(4, 0)
        print("bwd H=%d F=%d %s p=%.1f:  %s" % (h, f, str(dt)[6:], p, "   ".join(res)), flush=True)
