#!/usr/bin/env python3
"""csr_spmm kernel geometry at papers100M scale (features far beyond every cache): the variant hook
(cogdl_hip_csr_spmm_variant: VEC x LPR x UNROLL) and the XCD stripe on one eighth of the papers-shaped symmetrised graph
(13.9 M nodes, 4.0e8 edges, X = 7.1 GB), F = 128 fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
big = synth.papers100m_like(DEV, symmetrise=True, num_nodes=synth.PAPERS_NODES // 8, num_pairs=synth.PAPERS_PAIRS // 8)
g = synth.CSRGraph(big.rowptr.int(), big.colind, big.weight, big.num_nodes)
x = torch.randn(g.num_nodes, 128, device=DEV)
balg = g.nnz * 520 + g.num_nodes * 516
names = {-1: "auto (V2 L64 U8)", 0: "V4 L32 U8", 1: "V4 L32 U4", 3: "V2 L64 U8", 4: "V2 L64 U4", 7: "V4 L64 U8", 9: "V1 L64 U16", 10: "V2 L64 U16", 11: "V2 L32 U16", 15: "V2 L32 U12"}
for v, name in names.items():
    ms = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, v), 6, 2)
    print("variant %-18s %8.2f ms  %.0f GB/s (%.1f %% of 8 TB/s)" % (name, ms, balg / ms / 1e6, balg / ms / 1e6 / 80), flush=True)
for stripe in (0, 8, 32, 128, 1024):
    lib.cogdl_hip_set_tuning(0, stripe)
    ms = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x), 6, 2)
    print("xcd stripe %-5d %8.2f ms" % (stripe, ms), flush=True)
lib.cogdl_hip_set_tuning(0, 32)
