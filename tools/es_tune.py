#!/usr/bin/env python3
"""edge_softmax on the reddit-shaped graph: hub-row path variants (tuning keys 3 = long-row grid, 7 = lane width)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
n = 232_965
src, dst = synth.rmat_pairs(n, 57_300_000, seed=0, device=DEV)
g = synth.finalize(src, dst, n, norm=None)
del src, dst
deg = g.degrees()
thr = _lib.hip().cogdl_hip_long_row_threshold(g.nnz)
print("nnz=%d max_deg=%d thresh=%d edges in hub rows: %.1f%%" % (g.nnz, int(deg.max()), thr,
      100.0 * float(deg[deg > thr].sum()) / g.nnz), flush=True)
for h in (8, 4, 16):
    a = torch.randn(g.nnz, h, device=DEV)
    gr = torch.randn(g.nnz, h, device=DEV)
    sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
    for grid in (1024, 2040):
        for lanes in (0, 2):
            _lib.hip().cogdl_hip_set_tuning(3, grid)
            _lib.hip().cogdl_hip_set_tuning(7, lanes)
            f = timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a), 10)
            b = timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, gr), 10)
            print("H=%-3d long grid %4d  hub lanes %s   fwd %8.1f us (%5.0f GB/s)   bwd %8.1f us (%5.0f GB/s)" % (
                h, grid, "16B" if lanes == 0 else " 4B", f * 1e3, g.nnz * h * 8 / f / 1e6, b * 1e3,
                g.nnz * h * 12 / b / 1e6), flush=True)
    del a, gr, sm
_lib.hip().cogdl_hip_set_tuning(3, 1024)
_lib.hip().cogdl_hip_set_tuning(7, 0)
