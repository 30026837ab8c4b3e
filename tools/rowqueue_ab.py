#!/usr/bin/env python3
"""csr_spmm: the per-workgroup row queue (tuning key 16 = n: a workgroup owns n x as many consecutive rows, its waves pull
them from an LDS counter; 0 = the round-4 kernel with degree-sorted dealing) on the arxiv-sized uniform / R-MAT graphs and the
Reddit-shaped graph; kernel time by graph replay, bit-identity of the results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit_graph  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()


def run(name, g, cases, queues=(0, 1, 2, 4, 8, 16)):
    for f, dt in cases:
        x, w = torch.randn(g.n_cols, f, device=DEV).to(dt), g.weight.to(dt)
        res, ref = [], None
        same = True
        for q in queues:
            lib.cogdl_hip_set_tuning(16, q)
            out = csr_spmm_raw(g.rowptr, g.colind, w, x)
            ref = out if ref is None else ref
            same = same and torch.equal(out, ref)
            res.append("%d: %.1f" % (q, timeit_graph(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x)) * 1e3))
        lib.cogdl_hip_set_tuning(16, 0)
        print("%-14s F=%-3d %-8s us by queue depth | %s | bit-identical: %s" % (name, f, str(dt)[6:], "  ".join(res), same), flush=True)


CASES = ((128, torch.float32), (64, torch.float32), (40, torch.float32), (16, torch.float32), (64, torch.bfloat16), (128, torch.bfloat16))
run("arxiv-rmat", synth.arxiv_like(seed=0, topology="rmat").to(DEV), CASES)
run("arxiv-uniform", synth.arxiv_like(seed=0, topology="uniform").to(DEV), CASES)
run("reddit-like", synth.reddit_like(seed=0, device=DEV, norm="sym"), ((128, torch.float32), (64, torch.float32), (64, torch.bfloat16)), (0, 2, 8))
