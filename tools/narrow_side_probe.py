#!/usr/bin/env python3
"""install(narrow_side=True): the reference's unchanged GCNLayer (staged package) on the arxiv-shaped graph, a WIDENING
layer (128 -> 256, the first layer of examples/ogb/arxiv/gnn.py) forward + backward, in the reference's order
A (X W) and with the aggregation on the narrow side (A X) W."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import refpkg  # noqa: E402

refpkg.setup(install=True)
import torch  # noqa: E402

import cogdl_amd  # noqa: E402
import cogdl_amd.fused  # noqa: E402
from cogdl.data import Graph  # noqa: E402
from cogdl.layers import GCNLayer  # noqa: E402
from cogdl_amd import synth  # noqa: E402

DEV = "cuda:0"
g0 = synth.arxiv_like(seed=0)
deg = (g0.rowptr[1:] - g0.rowptr[:-1]).long()
row = torch.repeat_interleave(torch.arange(g0.num_nodes), deg)
g = Graph(edge_index=(row, g0.colind.long()), edge_weight=g0.weight, num_nodes=g0.num_nodes).to(DEV)
g.row_indptr  # build the CSR once
for fin, fout in ((128, 256), (256, 128)):
    torch.manual_seed(0)
    layer = GCNLayer(fin, fout).to(DEV)
    x = torch.randn(g0.num_nodes, fin, device=DEV, requires_grad=True)
    res = {}
    for mode in ("reference order", "narrow side"):
        if mode == "narrow side":
            cogdl_amd.install(narrow_side=True)
        for _ in range(5):
            layer(g, x).sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            out = layer(g, x)
            out.sum().backward()
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 30 * 1e3, out.detach())
        cogdl_amd.fused.uninstall_narrow_side()
    err = float((res["narrow side"][1] - res["reference order"][1]).abs().max() / res["reference order"][1].abs().max())
    print("GCNLayer %d -> %d fwd+bwd: reference order %.3f ms, narrow side %.3f ms  (max rel diff %.1e)" % (
        fin, fout, res["reference order"][0], res["narrow side"][0], err), flush=True)
