#!/usr/bin/env python3
"""Host-side neighbour sampler (BASELINE.json configs[3]: GraphSAGE on ogbn-products, fan-out [10,10]):
cogdl_amd.operators.sample.sample_adj_c vs the reference's own sampler.so (oracle/_ref, built from
cogdl/operators/sample/sample.cpp).  CPU only.  Usage: python tools/sampler_bench.py [nodes] [avg_degree]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.sample import sample_adj_c  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_449_029
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 50.5
t0 = time.time()
g = synth.scaled(n, deg, seed=0, topology="rmat", norm=None, self_loops=False)
indptr, indices = g.rowptr.long(), g.colind.long()
print("products-like graph: N=%d nnz=%d (%.1f s to build)" % (n, g.nnz, time.time() - t0), flush=True)

ref = None
try:
    from oracle import oracle

    if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), "_ref", "sampler.so")):
        ref = oracle.ref_sampler()
except Exception as e:  # the reference build only exists where /root/reference does
    print("reference sampler unavailable:", e)


def two_hop(fn, seeds, fanout):
    """What cogdl/data/sampler.py:NeighborSampler does per mini-batch: one sample_adj per layer, outermost first."""
    nodes = seeds
    edges = 0
    for k in fanout:
        rp, ci, nodes, eid = fn(indptr, indices, nodes, k, False)
        edges += ci.numel()
    return nodes.numel(), edges


for batch in (128, 1024):
    for name, fn in (("cogdl_amd", sample_adj_c), ("reference", None if ref is None else ref.sample_adj)):
        if fn is None:
            continue
        gen = torch.Generator().manual_seed(1)
        reps = 40 if batch == 128 else 10
        two_hop(fn, torch.randint(0, n, (batch,), generator=gen), [10, 10])
        t0 = time.perf_counter()
        tot_nodes = tot_edges = 0
        for _ in range(reps):
            a, b = two_hop(fn, torch.randint(0, n, (batch,), generator=gen), [10, 10])
            tot_nodes += a
            tot_edges += b
        dt = (time.perf_counter() - t0) / reps
        print("%-10s batch %5d fan-out [10,10]: %8.2f ms/batch  %6.2f M sampled edges/s  (%d nodes, %d edges per batch)" % (
            name, batch, dt * 1e3, tot_edges / reps / dt / 1e6, tot_nodes // reps, tot_edges // reps), flush=True)
