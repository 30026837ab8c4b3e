#!/usr/bin/env python3
"""Round 6: what the XCD-partitioned plan (cogdl_amd/xcdplan.py) is worth on the Reddit-shaped graph -- the kernels of
BASELINE configs[2], plan off / plan on, microseconds per call (plan build time reported once)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth, xcdplan  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func  # noqa: E402
from cogdl_amd.operators.spmm import SPMMFunction  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["off", "auto"]
dts = (torch.bfloat16, torch.float32)


def both(fn, label):
    res = []
    for mode in modes:
        xcdplan.MODE = mode
        fn()  # (builds the plans)
        torch.cuda.synchronize()
        res.append(timeit(fn, 10) * 1e3)
    print("%-44s " % label + "   ".join("%s %8.1f us" % (m, t) for m, t in zip(modes, res)), flush=True)


t0 = time.time()
for dt in dts:
    x = torch.randn(n, 64, device=dev).to(dt).requires_grad_()
    w = g.weight.to(dt)
    gout = torch.randn(n, 64, device=dev).to(dt)
    both(lambda: SPMMFunction.apply(g.rowptr, g.colind, x.detach(), w, False), "csr_spmm F=64 %s fwd" % str(dt)[6:])
    out = {}

    def fwd_bwd():
        o = SPMMFunction.apply(g.rowptr, g.colind, x, w, False)
        torch.autograd.grad(o, x, gout)
    both(fwd_bwd, "csr_spmm F=64 %s fwd+bwd" % str(dt)[6:])
for h, f in ((8, 8), (1, 41)):
    for dt in dts:
        ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
        ft = torch.randn(n, h, f, device=dev).to(dt).requires_grad_()
        grad = torch.randn(n, h, f, device=dev).to(dt)
        for p in (0.0, 0.5):
            both(lambda: fused_gat_dropout_func(ar.detach(), ac.detach(), g.rowptr, g.colind, 0.2, ft.detach(), p, seed=3),
                 "gat H=%d F=%d %s p=%.1f fwd" % (h, f, str(dt)[6:], p))

            def step():
                o = fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, p, seed=3)
                torch.autograd.grad(o, (ar, ac, ft), grad)
            both(step, "gat H=%d F=%d %s p=%.1f fwd+bwd" % (h, f, str(dt)[6:], p))
print("plans held: %.1f MB" % (xcdplan.XPLANS.bytes / 2 ** 20))
