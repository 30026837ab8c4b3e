#!/usr/bin/env python3
"""GPU experiment (not product): fused GAT forward, edge-wise online softmax (tuning key 5 = 1) vs chunk-wise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"


def run(g, tag, shapes):
    n, nnz = g.num_nodes, g.nnz
    for h, f in shapes:
        ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
        for dt in (torch.float32, torch.bfloat16):
            feat = torch.randn(n, h, f, device=DEV).to(dt)
            s = 4 if dt == torch.float32 else 2
            nbytes = nnz * (4 + h * 4 + h * f * s) + n * (4 + 2 * h * 4 + h * f * s)
            res = []
            for online in (1, 0):
                _lib.hip().cogdl_hip_set_tuning(5, online)
                ms = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), 10)
                res.append("%s %8.1f us (%4.1f%%)" % ("online" if online else "chunk ", ms * 1e3, nbytes / ms / 1e6 / 80))
            print("%s H=%d F=%d %s: %s" % (tag, h, f, str(dt)[6:], "   ".join(res)), flush=True)
    _lib.hip().cogdl_hip_set_tuning(5, 0)


SHAPES = [(1, 41), (1, 64), (1, 128), (2, 32), (4, 16), (8, 8), (4, 32), (8, 16), (8, 32)]
for topo in ("uniform", "rmat"):
    run(synth.arxiv_like(seed=0, topology=topo).to(DEV), "arxiv-" + topo, SHAPES)
n = 232_965
src, dst = synth.rmat_pairs(n, 57_300_000, seed=0, device=DEV)
g = synth.finalize(src, dst, n, norm=None)
del src, dst
run(g, "reddit-rmat", [(1, 41), (8, 8), (1, 64)])
