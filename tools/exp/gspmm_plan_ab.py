#!/usr/bin/env python3
"""Message operators over a shuffled COO list of the arxiv-sized graphs: ordinary launch vs the plan of the sorted view."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import synth, xcdplan  # noqa: E402
from cogdl_amd.operators import ops  # noqa: E402

dev = "cuda:0"


def us(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for topo in ("rmat",):
    gr = synth.arxiv_like(seed=0, topology=topo)
    deg = (gr.rowptr[1:] - gr.rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(gr.num_nodes), deg)
    sh = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(1))
    row, col = row[sh].to(dev), gr.colind.long()[sh].to(dev)
    g = types.SimpleNamespace(edge_index=(row, col), edge_weight=None)
    for f in (32, 64, 80, 96, 112, 128, 192, 256):
        x, ef = torch.randn(gr.num_nodes, f, device=dev), torch.randn(row.numel(), f, device=dev)
        line = "%-8s F=%-3d" % (topo, f)
        for mode in ("off", "force"):
            xcdplan.MODE = mode
            ops.clear_plans()
            line += "   %s: s_mul_e_sum %.1f us  scatter_add %.1f us" % (mode, us(lambda: ops.s_mul_e_sum(g, x, ef)), us(lambda: ops.scatter_add(ef, row, gr.num_nodes)))
            xg = x.clone().requires_grad_()

            def fb():
                xg.grad = None
                ops.s_mul_e_sum(g, xg, ef).sum().backward()

            line += "  fwd+bwd(x) %.1f us" % us(fb)
        print(line, flush=True)
