#!/usr/bin/env python3
"""Neighbour sampler (BASELINE.json configs[3]: GraphSAGE on ogbn-products, fan-out [10,10]):
cogdl_amd.operators.sample.sample_adj_c on the host (libcogdl_host.so) and, where a GPU is visible, on a GPU-resident
graph (cogdl_hip_sample_adj) vs the reference's own sampler.so (oracle/_ref, built from
cogdl/operators/sample/sample.cpp; only where /root/reference exists).  Usage: python tools/sampler_bench.py [nodes] [avg_degree]"""
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
GPU = torch.cuda.is_available()
if GPU:  # build on the GPU (seconds instead of half a minute), keep a host copy for the host operator
    src, dst = synth.rmat_pairs(n, int(n * deg / 2), 0, device="cuda:0")
    g = synth.finalize(src, dst, n, norm=None, self_loops=False)
    del src, dst
    indptr_d, indices_d = g.rowptr.long(), g.colind.long()
    indptr, indices = indptr_d.cpu(), indices_d.cpu()
else:
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


def two_hop(fn, seeds, fanout, on_gpu=False):
    """What cogdl/data/sampler.py:NeighborSampler does per mini-batch: one sample_adj per layer, outermost first."""
    nodes = seeds
    edges = 0
    for k in fanout:
        rp, ci, nodes, eid = fn(indptr_d, indices_d, nodes, k, False) if on_gpu else fn(indptr, indices, nodes, k, False)
        edges += ci.numel()
    return nodes.numel(), edges


def run(name, fn, batch):
    on_gpu = name.endswith("GPU")
    if on_gpu:
        gen = torch.Generator(device="cuda:0").manual_seed(1)  # seeds drawn on the GPU: a GPU-resident pipeline

        def seeds():
            return torch.randint(0, n, (batch,), generator=gen, device="cuda:0").unique()
        reps = 40
    else:
        gen = torch.Generator().manual_seed(1)

        def seeds():
            return torch.randint(0, n, (batch,), generator=gen)
        reps = 40 if batch == 128 else 10
    for _ in range(3 if on_gpu else 1):
        two_hop(fn, seeds(), [10, 10], on_gpu)
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot_nodes = tot_edges = 0
    for _ in range(reps):
        a, b = two_hop(fn, seeds(), [10, 10], on_gpu)
        tot_nodes += a
        tot_edges += b
    if on_gpu:
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-13s batch %5d fan-out [10,10]: %8.2f ms/batch  %7.2f M sampled edges/s  (%d nodes, %d edges per batch)" % (
        name, batch, dt * 1e3, tot_edges / reps / dt / 1e6, tot_nodes // reps, tot_edges // reps), flush=True)


BATCHES = (128, 1024, 8192)
if GPU:  # first: once the reference extension (a pybind torch module) has run in this process, GPU launches from it
    for batch in BATCHES:  # are several times slower -- an artifact of the checker, not of either sampler
        run("cogdl_amd GPU", sample_adj_c, batch)
for batch in BATCHES:
    run("cogdl_amd", sample_adj_c, batch)
    if ref is not None:
        run("reference", ref.sample_adj, batch)
