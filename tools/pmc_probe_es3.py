#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes of round 3 over the flat edge_softmax kernel only: the 1 GiB calibration
copy, then forward and backward, fp32 and bf16, H = 8, on the Reddit-shaped graph at its true size (3 launches each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402

dev = "cuda:0"
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
del a, b
g = synth.reddit_like(seed=0, device=dev)
for dt in (torch.float32, torch.bfloat16):
    v = torch.randn(g.nnz, 8, device=dev).to(dt)
    sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, v)
    for _ in range(3):
        es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, v)
    for _ in range(3):
        es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, v)
    torch.cuda.synchronize()
    print("done", dt, flush=True)
    del v, sm
