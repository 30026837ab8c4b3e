#!/usr/bin/env python3
"""Workload for the GCN-epoch kernel breakdown (not part of the product): EPOCHS training steps of bench.py's GCN with
the MFMA linear kernels, nothing else.  Run under `rocprofv3 --kernel-trace --stats`; tools/epoch_breakdown.py folds
the kernel statistics into per-epoch categories."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cogdl_amd import synth  # noqa: E402

EPOCHS, WARMUP = 40, 5  # tools/epoch_breakdown.py divides by EPOCHS + WARMUP

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    g = synth.arxiv_like(seed=0, topology="uniform")
    gd = g.to(dev)
    x = torch.randn(g.num_nodes, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    r = bench.gcn_epoch_ms(gd, gd.rowptr.long(), gd.colind.long(), x, reps=EPOCHS, warmup=WARMUP, mfma_linear=True)
    print("epoch ms (unprofiled clock inside the profiled run): %.3f  min %.3f" % (r["ms"], r["min_ms"]))
