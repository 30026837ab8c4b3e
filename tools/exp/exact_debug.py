import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan, _lib
from cogdl_amd.operators.spmm import csr_spmm_raw, csr_spmm_xcd_raw
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
bound = int(_lib.hip().cogdl_hip_exact_row_edges(g.nnz))
deg = g.degrees().to(dev)
for k in (32, 64):
    x = torch.randn(n, k, device=dev)
    ref = csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
    for split in (64, bound):
        plan = xcdplan.build(g.rowptr, g.colind, split=split)
        out = csr_spmm_xcd_raw(plan, g.weight, x)
        bad = (out != ref).any(dim=1)
        short = deg <= split
        bs = bad & short
        print("k", k, "split", split, "rows", n, "short", int(short.sum()), "bad short rows", int(bs.sum()), "bad long", int((bad & ~short).sum()),
              "deg of bad short: min %s max %s" % ((int(deg[bs].min()), int(deg[bs].max())) if bs.any() else (None, None)),
              "max abs diff short %.3e" % float((out - ref)[short].abs().max()))
        if bs.any():
            r = int(torch.nonzero(bs)[0]); print("   first bad row", r, "deg", int(deg[r]), out[r, :4].tolist(), ref[r, :4].tolist())
