"""Would a degree-ordered row schedule help csr_spmm where X is far beyond every cache?  One eighth of the papers100M-shaped
symmetrised graph (13.9 M rows, 4.0e8 edges, X = 7.1 GB at F = 128 fp32), rows in id order against the SAME rows physically
re-ordered by decreasing degree (what a row-order indirection in the row blocks would walk), and against rows ordered in
degree BUCKETS of a window (locality of the row pointer / output kept within 64 K-row windows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth
from cogdl_amd.operators.spmm import csr_spmm_raw
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.papers100m_like(dev, True, num_nodes=synth.PAPERS_NODES // 8, num_pairs=synth.PAPERS_PAIRS // 8)
n, nnz = g.num_nodes, g.nnz
rp = g.rowptr
deg = rp[1:] - rp[:-1]
print("rows %d edges %d max degree %d" % (n, nnz, int(deg.max())), flush=True)

def reorder(order):
    nd = deg[order]
    nrp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nd, 0, out=nrp[1:])
    src = torch.repeat_interleave(rp[:-1][order] - nrp[:-1], nd) + torch.arange(nnz, device=dev)
    return nrp.int(), g.colind[src].contiguous(), g.weight[src].contiguous()

variants = {"id order": (rp.int(), g.colind, g.weight)}
variants["by degree (descending)"] = reorder(torch.argsort(deg, descending=True, stable=True))
win = 1 << 16
key = (torch.arange(n, device=dev) // win) * (int(deg.max()) + 1) + (int(deg.max()) - deg)
variants["by degree inside 64 K-row windows"] = reorder(torch.argsort(key, stable=True))
del key
for f in (128, 64):
    x = torch.randn(n, f, device=dev)
    for name, (a, b, c) in variants.items():
        t = timeit(lambda: csr_spmm_raw(a, b, c, x), 5) 
        alg = nnz * (8 + 4 * f) + n * (4 + 4 * f)
        print("F=%-4d %-36s %8.2f ms   %.3f of 8 TB/s" % (f, name, t, alg / (t * 1e-3) / 8e12), flush=True)
