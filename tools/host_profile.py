#!/usr/bin/env python3
"""Host-side cost of one bench step (csrspmm fwd + bwd through autograd): cProfile over the step loop.
Usage on the GPU box: python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.spmm import csrspmm  # noqa: E402

dev = "cuda:0"
g = synth.arxiv_like(seed=0).to(dev)
r64, c64 = g.rowptr.long(), g.colind.long()
x = torch.randn(g.num_nodes, 128, device=dev, requires_grad=True)
gout = torch.randn(g.num_nodes, 128, device=dev)


def step():
    out = csrspmm(r64.int(), c64.int(), x, g.weight, True)
    x.grad = None
    out.backward(gout)


for _ in range(20):
    step()
torch.cuda.synchronize()
# host-only time per step: enqueue 200 steps without waiting for the GPU
t0 = time.perf_counter()
for _ in range(200):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.1f us/step, wall %.1f us/step" % (t_host / 200 * 1e6, t_all / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
