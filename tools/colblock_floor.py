#!/usr/bin/env python3
"""Companion of colblock_probe.py: what would PERFECT column locality buy?  The same Reddit-shaped launches with every column
id folded into a 1024-row window (the gathered table is 128 / 256 KB: every gather an L2 hit) -- the engine's issue / latency
floor on this degree distribution -- and the virtual-row launch (B = 2, 4, 8) on the true and on the folded ids."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402
from tools.colblock_probe import virtual_rows, DEV  # noqa: E402

g = synth.reddit_like(seed=0, device=DEV, norm="sym")
m = g.num_nodes
fold = (g.colind & 1023).contiguous()
fold64k = (g.colind & 16383).contiguous()
for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    x = torch.randn(m, 64, device=DEV).to(dt)
    w = g.weight.to(dt)
    t0 = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10)
    t1 = timeit(lambda: csr_spmm_raw(g.rowptr, fold, w, x), 10)
    t2 = timeit(lambda: csr_spmm_raw(g.rowptr, fold64k, w, x), 10)
    print("%s plain %8.1f us   columns folded to 1024 rows %8.1f us   to 16384 rows (%.1f MB) %8.1f us" % (
        name, t0 * 1e3, t1 * 1e3, 16384 * 64 * x.element_size() / 1e6, t2 * 1e3), flush=True)
    for B in (2, 4, 8):
        for T in (256, 1024):
            order, vrowptr, vrow, vblk = virtual_rows(g.rowptr, g.colind, m, B, T)
            ci, wv, rp = g.colind[order].contiguous(), w[order].contiguous(), vrowptr.int()
            tv = timeit(lambda: csr_spmm_raw(rp, ci, wv, x), 10)
            cf = (ci & 1023).contiguous()
            tf = timeit(lambda: csr_spmm_raw(rp, cf, wv, x), 10)
            print("   virt B=%d T=%-4d V=%8d   true ids %8.1f us   folded ids %8.1f us" % (B, T, vrow.numel(), tv * 1e3, tf * 1e3), flush=True)
