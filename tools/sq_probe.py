#!/usr/bin/env python3
"""Workload for one `rocprofv3 --pmc SQ_...` pass: what bounds the row-reduce kernels on the Reddit-shaped graph -- issue
(SQ_ACTIVE_INST_*), issue stalls (SQ_WAIT_INST_ANY) or parked waves (SQ_WAIT_ANY = s_waitcnt / barriers)?  csr_spmm F = 64
bf16 / fp32 with the true and with FOLDED column ids (every gather an L2 hit), fused GAT forward bf16 H = 8 x F = 8.
tools/sq_summarize.py folds the CSV."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402

dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "reddit"
if which == "reddit":
    g = synth.reddit_like(seed=0, device=dev, norm="sym")
else:
    g = synth.arxiv_like(seed=0, topology="rmat").to(dev)
n = g.num_nodes
fold = (g.colind & 1023).contiguous()
for dt in (torch.bfloat16, torch.float32):
    x, w = torch.randn(n, 64, device=dev).to(dt), g.weight.to(dt)
    for ci in (g.colind, fold):
        for _ in range(2):
            csr_spmm_raw(g.rowptr, ci, w, x)
        torch.cuda.synchronize()
ar, ac = torch.randn(n, 8, device=dev), torch.randn(n, 8, device=dev)
feat = torch.randn(n, 8, 8, device=dev).bfloat16()
for _ in range(2):
    gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat)
torch.cuda.synchronize()
print("done")
