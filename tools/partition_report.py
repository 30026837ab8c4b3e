#!/usr/bin/env python3
"""What the GPU partitioner (cogdl_amd.dist.partition) buys on this box: halo rows and remote edges per rank before and
after the reordering, 8 ranks -- on the ogbn-products-shaped R-MAT graph (2.45 M vertices, 1.2e8 edges: structureless
apart from its degree distribution) and on a graph with real locality hidden behind a random relabelling (a ring
lattice of 2 M vertices, 40 neighbours each, plus 200 long links), on the same lattice with 100,000 long links (where
breadth-first levels interleave distant regions) and on 64 planted communities -- orders: none / degree / bfs / multilevel
(cogdl_amd/partitioner.py).  Times are wall times of partition() itself."""
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
    for order in ("none", "degree", "bfs", "multilevel"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        part = partition(rp, ci, WORLD, order=order)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows = [int(part.bounds[p + 1] - part.bounds[p]) for p in range(WORLD)]
        rem = sum(e for e, _ in part.halo_after)
        print("%-26s order=%-10s  %6.2f s  remote edges %5.1f %%  halo rows per rank: max %9d (%.2f x its rows)  mean %9d" % (
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
del g
# the same lattice with MANY long links (5 % of the vertices get one): breadth-first levels interleave distant regions
src = torch.cat([base.repeat(hw), torch.randint(0, n, (n // 20,), generator=gen, device=DEV)])
dst = torch.cat([torch.cat([(base + d) % n for d in range(1, hw + 1)]), torch.randint(0, n, (n // 20,), generator=gen, device=DEV)])
g = synth.finalize(shuffle[src], shuffle[dst], n, norm=None, self_loops=False)
report("ring lattice, 5% long links", g.rowptr.long(), g.colind.long())
del g
# planted communities: 64 of them (8 per rank), 20 neighbours inside, 2 outside, ids shuffled
k, size = 64, 30000
n = k * size
comm = torch.arange(n, device=DEV) // size
src_in = torch.arange(n, device=DEV).repeat_interleave(10)
dst_in = comm[src_in] * size + torch.randint(0, size, (src_in.numel(),), generator=gen, device=DEV)
src_out = torch.arange(n, device=DEV)
dst_out = torch.randint(0, n, (n,), generator=gen, device=DEV)
shuffle = torch.randperm(n, generator=gen, device=DEV)
g = synth.finalize(shuffle[torch.cat([src_in, src_out])], shuffle[torch.cat([dst_in, dst_out])], n, norm=None, self_loops=False)
report("64 planted communities", g.rowptr.long(), g.colind.long())
