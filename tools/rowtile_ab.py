#!/usr/bin/env python3
"""A/B of csr_spmm's row tiles (csrc/rowreduce.h: rowreduce_tile_kernel; tuning key 14: 1 = off, 2 = on): the arxiv-sized
graphs (uniform / R-MAT), every width CogDL's gcn runs, fp32 / bf16, weighted / unweighted; rows of exactly d edges."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()


def ab(fn):
    out = []
    for v in (1, 2):
        lib.cogdl_hip_set_tuning(14, v)
        out.append(timeit(fn, 30) * 1e3)
    lib.cogdl_hip_set_tuning(14, 0)
    return out


for topo in ("rmat", "uniform"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    n, nnz = g.num_nodes, g.nnz
    for f, dt in ((16, torch.float32), (40, torch.float32), (64, torch.float32), (128, torch.float32), (256, torch.float32),
                  (64, torch.bfloat16), (128, torch.bfloat16)):
        x, w = torch.randn(n, f, device=DEV).to(dt), g.weight.to(dt)
        s = x.element_size()
        off, on = ab(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x))
        nb = nnz * (4 + s + f * s) + n * (4 + f * s)
        print("arxiv-%-7s F=%-3d %-8s  plain %7.1f us (%4.1f %%)   row tiles %7.1f us (%4.1f %%)" % (
            topo, f, str(dt)[6:], off, nb / (off * 1e-6) / 8e12 * 100, on, nb / (on * 1e-6) / 8e12 * 100), flush=True)
n = 169_343
x = torch.randn(n, 64, device=DEV)
res = []
for d in (1, 2, 4, 8, 15, 32):
    rowptr = (torch.arange(n + 1, device=DEV) * d).int()
    colind = torch.randint(0, n, (n * d,), device=DEV).int()
    w = torch.rand(n * d, device=DEV)
    off, on = ab(lambda: csr_spmm_raw(rowptr, colind, w, x))
    res.append("d=%-2d %5.1f -> %5.1f us" % (d, off, on))
print("rows of exactly d edges, F=64 f32:  " + "   ".join(res))
