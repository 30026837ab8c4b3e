#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over the HBM-RESIDENT csr_spmm: the 1 GiB calibration copy, then three launches
over one papers100M-sized shard's local block (13.9 M rows, 4.1e8 edges, X = 7.1 GB: far beyond L2 + Infinity Cache) --
the same shard bench.py's weak_scaling_base / roofline.hbm_resident times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd.dist import _papers_like_shard  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402

dev = torch.device("cuda:0")
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
del a, b
n = 111_059_956 // 8
rowptr, cols, w = _papers_like_shard(0, 1, n, 28.8, 0.1, 0, dev, 0.25)
rp, ci = rowptr.int(), cols.int()
del rowptr, cols
x = torch.randn(n, 128, device=dev)
for _ in range(3):
    csr_spmm_raw(rp, ci, w, x)
torch.cuda.synchronize()
print("shard nnz", ci.numel(), "algorithmic bytes", ci.numel() * (8 + 512) + n * (4 + 512))
