#!/usr/bin/env python3
"""Does a plan-time row order (rows by decreasing degree) help the ORDINARY csr_spmm launch on the arxiv-sized R-MAT graph?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402

dev = "cuda:0"


def us(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for topo in ("rmat", "uniform"):
    g = synth.arxiv_like(seed=0, topology=topo).to(dev)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    m = deg.numel()
    orders = {"none": None, "desc": torch.argsort(deg, descending=True, stable=True).int(),
              "asc": torch.argsort(deg, stable=True).int()}
    for win in (4096, 65536):
        key = (torch.arange(m, device=dev) // win) * (int(deg.max()) + 1) + (int(deg.max()) - deg)
        orders["desc/%d" % win] = torch.argsort(key, stable=True).int()
    for f in (128, 64, 40):
        x = torch.randn(m, f, device=dev)
        ref = csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
        line = "%-8s F=%-3d" % (topo, f)
        for name, o in orders.items():
            t = us(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, row_order=o))
            assert torch.equal(csr_spmm_raw(g.rowptr, g.colind, g.weight, x, row_order=o), ref)
            line += "  %s %.1f us" % (name, t)
        print(line, flush=True)
