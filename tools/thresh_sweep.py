#!/usr/bin/env python3
"""Long-row threshold sweep (tuning key 1) of csr_spmm on the arxiv-sized R-MAT graph, the Reddit-shaped graph and a
papers100M-shaped row segment: is the automatic rule (pick_long_thresh, csrc/rowreduce.h) still where the minimum is?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()


def sweep(name, g, widths, threshes):
    for f, dt in widths:
        x, w = torch.randn(g.n_cols, f, device=DEV).to(dt), g.weight.to(dt)
        auto = lib.cogdl_hip_long_row_threshold(g.nnz)
        res = []
        for t in threshes:
            lib.cogdl_hip_set_tuning(1, t)
            res.append("%d: %.1f" % (t, timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10) * 1e3))
        lib.cogdl_hip_set_tuning(1, 0)
        base = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10) * 1e3
        print("%-14s F=%-3d %-8s auto(%d) %.1f us | %s" % (name, f, str(dt)[6:], auto, base, "  ".join(res)), flush=True)
        del x, w


W = ((128, torch.float32), (64, torch.float32), (40, torch.float32), (64, torch.bfloat16), (128, torch.bfloat16))
g = synth.arxiv_like(seed=0, topology="rmat").to(DEV)
sweep("arxiv-rmat", g, W, (32, 64, 128, 256, 512))
g = synth.reddit_like(seed=0, device=DEV, norm="sym")
sweep("reddit-like", g, W, (128, 256, 512, 1024, 2048))
del g
torch.cuda.empty_cache()
# one ~2^28-edge row range of the papers-shaped symmetrised graph (a quarter of the nodes keeps the degree law)
big = synth.papers100m_like(DEV, symmetrise=True, num_nodes=synth.PAPERS_NODES // 8, num_pairs=synth.PAPERS_PAIRS // 8)
m = big.num_nodes
g32 = synth.CSRGraph(big.rowptr.int(), big.colind, big.weight, m)
sweep("papers/8 sym", g32, ((128, torch.float32), (64, torch.float32)), (256, 512, 1024, 2048))
