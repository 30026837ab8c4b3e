#!/usr/bin/env python3
"""Host-side (Python) cost of one GCN training epoch of bench.py: cProfile over 50 steps, cumulative time per function.
The epoch is launch-bound (~100 kernel launches, GPU busy 1.17 of 1.29 ms): this shows where the host time goes."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import linear as cogdl_linear, synth  # noqa: E402
from cogdl_amd.operators.spmm import csrspmm  # noqa: E402

dev = torch.device("cuda:0")
g = synth.arxiv_like(seed=0).to(dev)
r64, c64 = g.rowptr.long(), g.colind.long()
n = g.num_nodes
x = torch.randn(n, 128, device=dev)
cogdl_linear.install()
lin1, lin2 = torch.nn.Linear(128, 64).to(dev), torch.nn.Linear(64, 40).to(dev)
drop = torch.nn.Dropout(0.5)
opt = torch.optim.Adam(list(lin1.parameters()) + list(lin2.parameters()), lr=0.01, weight_decay=5e-4)
y = torch.randint(0, 40, (n,), device=dev)
mask = torch.rand(n, device=dev) < 0.537


def step():
    opt.zero_grad(set_to_none=True)
    h = csrspmm(r64.int(), c64.int(), lin1(x), g.weight, True)
    h = drop(torch.relu_(h))
    out = csrspmm(r64.int(), c64.int(), lin2(h), g.weight, True)
    torch.nn.functional.cross_entropy(out[mask], y[mask]).backward()
    opt.step()


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.1f us/epoch, wall %.1f us/epoch" % (t_host / 100 * 1e6, t_all / 100 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(35)
