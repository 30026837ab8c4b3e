#!/usr/bin/env python3
"""Fused GAT forward / backward on the arxiv-sized graphs (2.4 M edges: a launch is ~100-400 us, bound by latency and by
how evenly the rows fill the waves): kernel choice (tuning key 5) x vector cap (key 4), H=8 x F=8 and H=1 x F=48."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func, gat_forward  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
for topo in ("rmat", "uniform"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    n, nnz = g.num_nodes, g.nnz
    for h, f, dt in ((8, 8, torch.float32), (8, 8, torch.bfloat16), (1, 48, torch.bfloat16), (4, 32, torch.float32)):
        s = 2 if dt == torch.bfloat16 else 4
        ar, ac = torch.randn(n, h, device=DEV).requires_grad_(), torch.randn(n, h, device=DEV).requires_grad_()
        feat = torch.randn(n, h, f, device=DEV).to(dt).requires_grad_()
        gout = torch.randn(n, h, f, device=DEV).to(dt)
        b_fwd = nnz * (4 + 4 * h + h * f * s) + n * (4 + 8 * h + h * f * s)
        res = []
        for kern in (1, 2):
            for vcap in (0, 4, 2):
                lib.cogdl_hip_set_tuning(5, kern)
                lib.cogdl_hip_set_tuning(4, vcap)
                with torch.no_grad():
                    us = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), 20) * 1e3
                res.append("%s/v%s %5.0f us (%2.0f%%)" % ("on" if kern == 1 else "ch", vcap or "max", us, b_fwd / (us * 1e-6) / 8e12 * 100))
        lib.cogdl_hip_set_tuning(5, 0)
        bres = []
        for vcap in (0, 4, 2):
            lib.cogdl_hip_set_tuning(4, vcap)
            o = fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, feat, 0.0)
            us = timeit(lambda: torch.autograd.grad(o, (ar, ac, feat), gout, retain_graph=True), 20) * 1e3
            bres.append("v%s %5.0f us" % (vcap or "max", us))
            del o
        lib.cogdl_hip_set_tuning(4, 0)
        print("arxiv-%-7s H=%d F=%-2d %-8s fwd: %s | bwd: %s" % (topo, h, f, str(dt)[6:], "  ".join(res), "  ".join(bres)), flush=True)
