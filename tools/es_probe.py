#!/usr/bin/env python3
"""Where does the flat edge_softmax kernel's time go?  Timing-only variants (tuning key 9: WRONG results)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
g = synth.reddit_like(seed=0, device=DEV)
for h in (8, 1):
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(g.nnz, h, device=DEV).to(dt)
        sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
        s = a.element_size()
        variants = ((0, "default"), (1, "no exchange"), (8, "2 exp / elem"), (9, "2 exp no exch"))
        if dt != torch.float32:  # 16-bit values: one element per LDS access (bit 4); half-size tiles at 6 workgroups per CU (bit 2)
            variants += ((16, "1 elem / LDS op"), (4, "8k tiles x6"))
        for dbg, label in variants:
            lib.cogdl_hip_set_tuning(9, dbg)
            f = timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a), 10)
            b = timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, a), 10)
            print("reddit H=%d %-8s %-13s fwd %8.1f us (%5.0f GB/s)  bwd %8.1f us (%5.0f GB/s)" % (
                h, str(dt)[6:], label, f * 1e3, g.nnz * h * 2 * s / f / 1e6, b * 1e3, g.nnz * h * 3 * s / b / 1e6), flush=True)
        lib.cogdl_hip_set_tuning(9, 0)
