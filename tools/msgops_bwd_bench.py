#!/usr/bin/env python3
"""s_mul_e_sum / s_add_e_mean forward + backward at the arxiv-shaped size (F = 64): this library's fused operators (gspmm
forward; backward = gspmm over the source-sorted view + cogdl_hip_gspmm_edge_grad) against the reference's torch composition
(cogdl/operators/ops.py:43-52: gather, multiply, scatter_add_ and autograd through them) on the same GPU -- time per step
and peak memory beyond the inputs."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd.operators import ops  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
n, e, k = 169_343, 2_332_486, 64
gen = torch.Generator(device=DEV).manual_seed(0)
row = torch.sort(torch.randint(0, n, (e,), device=DEV, generator=gen)).values
col = torch.randint(0, n, (e,), device=DEV, generator=gen)
w = torch.rand(e, device=DEV, generator=gen)
G = torch.randn(n, k, device=DEV, generator=gen)
g = types.SimpleNamespace(edge_index=(row, col), edge_weight=w)


def reference(op1, op2, x, ef, weight):
    msg = {"mul": torch.mul, "add": torch.add}[op1](x[col], ef)
    if weight:
        msg = msg * w.view(-1, 1)
    out = torch.zeros(n, k, device=DEV).scatter_add_(0, row.view(-1, 1).expand(e, k), msg)
    if op2 == "mean":
        deg = torch.zeros(n, device=DEV).scatter_add_(0, row, torch.ones(e, device=DEV))
        inv = deg.pow(-1)
        inv[torch.isinf(inv)] = 0
        out = out * inv.view(-1, 1)
    return out


for op1, op2, weight, ef_grad in (("mul", "sum", False, False), ("mul", "sum", True, True), ("add", "mean", True, True)):
    x = torch.randn(n, k, device=DEV, generator=gen).requires_grad_()
    ef = torch.randn(e, k, device=DEV, generator=gen).requires_grad_(ef_grad)
    fused = getattr(ops, "s_%s_e_%s" % (op1, op2))

    def step_fused():
        x.grad = None
        ef.grad = None
        fused(g, x, ef, weight=weight).backward(G)

    def step_ref():
        x.grad = None
        ef.grad = None
        reference(op1, op2, x, ef, weight).backward(G)

    res = []
    for name, fn in (("fused", step_fused), ("torch composition", step_ref)):
        fn()
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        ms = timeit(fn, 10)
        res.append("%s %7.3f ms, peak +%6.1f MB" % (name, ms, (torch.cuda.max_memory_allocated() - base) / 1e6))
    print("s_%s_e_%s weight=%-5s grad(e_feat)=%-5s [E,F] = %.0f MB | %s" % (op1, op2, weight, ef_grad, e * k * 4 / 1e6, "   ".join(res)), flush=True)
