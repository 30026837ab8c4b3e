#!/usr/bin/env python3
"""Where does csr_spmm's time go on the power-law graph?  The row blocks alone, the long-row workgroups alone (tuning key 13,
timing only) and both, on the arxiv-sized R-MAT graph and the uniform one; F = 64 / 128 fp32, F = 64 bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
for topo in ("rmat", "uniform"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    for f, dt in ((64, torch.float32), (128, torch.float32), (40, torch.float32), (64, torch.bfloat16)):
        x, w = torch.randn(g.num_nodes, f, device=DEV).to(dt), g.weight.to(dt)
        res = []
        for dbg in (0, 1, 2):
            lib.cogdl_hip_set_tuning(13, dbg)
            res.append(timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 30) * 1e3)
        lib.cogdl_hip_set_tuning(13, 0)
        print("arxiv-%-7s F=%-3d %-8s  both %7.1f us   long-row workgroups alone %7.1f us   row blocks alone %7.1f us" % (
            topo, f, str(dt)[6:], res[0], res[1], res[2]), flush=True)

# what bounds the ROW BLOCKS on the R-MAT graph (half its rows have <= 2 edges)?  rows alone, with / without the degree-sorted
# dealing (key 2), the wave-scope split (key 12), the XCD stripe (key 0)
g = synth.arxiv_like(seed=0, topology="rmat").to(DEV)
x, w = torch.randn(g.num_nodes, 64, device=DEV), g.weight
lib.cogdl_hip_set_tuning(13, 2)
for name, sets in (("default", {}), ("no row dealing", {2: 1}), ("wave split 32", {12: 32}), ("wave split 16", {12: 16}),
                   ("xcd stripe 0", {0: 0}), ("xcd stripe 8", {0: 8}), ("xcd stripe 128", {0: 128})):
    for k, v in sets.items():
        lib.cogdl_hip_set_tuning(k, v)
    ms = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 30) * 1e3
    for k in sets:
        lib.cogdl_hip_set_tuning(k, {0: 32}.get(k, 0))
    print("rmat F=64 row blocks alone, %-16s %7.1f us" % (name, ms), flush=True)
lib.cogdl_hip_set_tuning(13, 0)
# rows sorted by degree (a permuted graph: the same work, perfectly balanced waves) -- how much is imbalance worth at all?
deg = g.degrees()
order = torch.argsort(deg, stable=True)
from cogdl_amd.dist import permute_graph  # noqa: E402
rp2, ci2, w2 = permute_graph(g.rowptr.long(), g.colind.long(), g.weight, order)
rp2, ci2 = rp2.int(), ci2.int()
for dbg, name in ((0, "both"), (1, "long-row workgroups alone"), (2, "row blocks alone")):
    lib.cogdl_hip_set_tuning(13, dbg)
    ms = timeit(lambda: csr_spmm_raw(rp2, ci2, w2, x), 30) * 1e3
    print("rmat F=64, rows AND columns relabelled by ascending degree, %-26s %7.1f us" % (name, ms), flush=True)
lib.cogdl_hip_set_tuning(13, 0)
