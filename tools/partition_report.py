#!/usr/bin/env python3
"""What the GPU partitioner (cogdl_amd.dist.partition) buys on this box: halo rows and remote edges per rank before and
after the reordering, 8 ranks -- on the ogbn-products-shaped R-MAT graph (2.45 M vertices, 1.2e8 edges: structureless
apart from its degree distribution) and on a graph with real locality hidden behind a random relabelling (a ring
lattice of 2 M vertices, 40 neighbours each, plus 200 long links).  Times are wall times of partition() itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.dist import partition  # noqa: E402

DEV = "cuda:0"
WORLD = 8


def report(name, rp, ci):
    n, nnz = rp.numel() - 1, ci.numel()
    for order in ("none", "degree", "bfs"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        part = partition(rp, ci, WORLD, order=order)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows = [int(part.bounds[p + 1] - part.bounds[p]) for p in range(WORLD)]
        rem = sum(e for e, _ in part.halo_after)
        print("%-22s order=%-6s  %6.2f s  remote edges %5.1f %%  halo rows per rank: max %9d (%.2f x its rows)  mean %9d" % (
            name, order, dt, 100.0 * rem / nnz, max(h for _, h in part.halo_after), part.halo_fraction(),
            sum(h for _, h in part.halo_after) // WORLD), flush=True)
        del part


n = 2_449_029
src, dst = synth.rmat_pairs(n, int(n * 50.5 / 2), 0, device=DEV)
g = synth.finalize(src, dst, n, norm=None, self_loops=False)
del src, dst
report("products-shaped R-MAT", g.rowptr.long(), g.colind.long())
del g
n = 2_000_000
gen = torch.Generator(device=DEV).manual_seed(1)
base = torch.arange(n, device=DEV)
hw = 20
src = torch.cat([base.repeat(hw), torch.randint(0, n, (n // 10000,), generator=gen, device=DEV)])
dst = torch.cat([torch.cat([(base + d) % n for d in range(1, hw + 1)]), torch.randint(0, n, (n // 10000,), generator=gen, device=DEV)])
shuffle = torch.randperm(n, generator=gen, device=DEV)
g = synth.finalize(shuffle[src], shuffle[dst], n, norm=None, self_loops=False)
report("hidden ring lattice", g.rowptr.long(), g.colind.long())
