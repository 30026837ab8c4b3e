#!/usr/bin/env python3
"""Host sampler (libcogdl_host, cogdl_host_sample_adj_mt) on THIS machine's host cores: two-hop batches on the
products-shaped graph for several OpenMP thread counts.  The graph is built on the GPU when there is one (seconds)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.sample import sample_adj_c  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_449_029
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
src, dst = synth.rmat_pairs(n, int(n * 50.5 / 2), 0, device=dev)
g = synth.finalize(src, dst, n, norm=None, self_loops=False)
indptr, indices = g.rowptr.long().cpu(), g.colind.long().cpu()
del src, dst, g
print("torch threads", torch.get_num_threads(), "cores", os.cpu_count(), flush=True)
for batch in (1024, 8192):
    for threads in (1, 2, 4, 8, 16):
        os.environ["COGDL_AMD_SAMPLER_THREADS"] = str(threads)
        gen = torch.Generator().manual_seed(1)
        ts = []
        for rep in range(12):
            seeds = torch.randint(0, n, (batch,), generator=gen).unique()
            t0 = time.perf_counter()
            nodes = seeds
            for k in (10, 10):
                _, _, nodes, _ = sample_adj_c(indptr, indices, nodes, k, False)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("batch %5d threads %2d: median %8.2f ms  min %8.2f ms" % (batch, threads, ts[len(ts) // 2] * 1e3, ts[0] * 1e3), flush=True)
