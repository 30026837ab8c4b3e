#!/usr/bin/env python3
"""csr_spmm: cap on the number of long-row workgroups (tuning key 3; default 1024) on the arxiv-sized R-MAT graph and the
Reddit-shaped graph -- a long-row workgroup walks its run of chunks as a serial chain, more workgroups = shorter chains."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit_graph  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
for name, g in (("arxiv-rmat", synth.arxiv_like(seed=0, topology="rmat").to(DEV)), ("reddit-like", synth.reddit_like(seed=0, device=DEV, norm="sym"))):
    for f, dt in ((128, torch.float32), (64, torch.float32), (40, torch.float32), (64, torch.bfloat16)):
        x, w = torch.randn(g.n_cols, f, device=DEV).to(dt), g.weight.to(dt)
        res = []
        for cap in (256, 512, 1024, 2040):
            lib.cogdl_hip_set_tuning(3, cap)
            _lib._WS_BYTES.clear()
            res.append("%d: %.1f" % (cap, timeit_graph(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x)) * 1e3))
        lib.cogdl_hip_set_tuning(3, 1024)
        print("%-12s F=%-3d %-8s us by long-row workgroup cap | %s" % (name, f, str(dt)[6:], "  ".join(res)), flush=True)
