#!/usr/bin/env python3
"""Workload for PMC passes over the dense-side kernels: linear_fwd (two GCN shapes) and linear_wgrad, 20 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import linear as L  # noqa: E402

dev = "cuda:0"
n = 169_343
for k, o in ((128, 64), (64, 40)):
    x = torch.randn(n, k, device=dev)
    w = torch.randn(o, k, device=dev)
    b = torch.randn(o, device=dev)
    g = torch.randn(n, o, device=dev)
    for _ in range(20):
        y = L.tall_skinny_matmul(x, w, b, True)
    for _ in range(20):
        L.linear_wgrad(x, g)
    torch.cuda.synchronize()
print("ok")
