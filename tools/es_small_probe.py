#!/usr/bin/env python3
"""Kernel-trace workload: the flat edge_softmax kernels on the arxiv-shaped graphs (small: launch- and latency-bound)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402

for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to("cuda:0")
    for h in (1, 8):
        a = torch.randn(g.nnz, h, device="cuda:0")
        s = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
        for _ in range(5):
            es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
        for _ in range(5):
            es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, s, a)
        torch.cuda.synchronize()
        print(topo, h, "done", flush=True)
