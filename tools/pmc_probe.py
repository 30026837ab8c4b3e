#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a calibration copy/read of known size (FETCH_SIZE/WRITE_SIZE
are mis-scaled on gfx950, MI355X_MICROARCH.md section HBM) followed by the bench's csr_spmm launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402

dev = "cuda:0"
# calibration: 1 GiB float32 copy (1 GiB read + 1 GiB written, far beyond the 256 MiB Infinity Cache)
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
for _ in range(3):
    a.sum()
torch.cuda.synchronize()
del a, b
for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to(dev)
    x = torch.randn(g.num_nodes, 128, device=dev)
    for _ in range(10):
        csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
    torch.cuda.synchronize()
# a graph whose feature matrix (2 GiB) cannot live in the Infinity Cache: true HBM behaviour
g = synth.scaled(4_000_000, 15, seed=1).to(dev)
x = torch.randn(g.num_nodes, 128, device=dev)
for _ in range(5):
    csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
torch.cuda.synchronize()
print("probe done: big graph nnz", g.nnz)
