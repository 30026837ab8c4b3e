#!/usr/bin/env python3
"""GPU experiment (not product): fused GAT forward under the tuning knobs (vector width, fast exp)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.fused_gat import gat_forward  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"


def run(g, tag, h, f):
    n, nnz = g.num_nodes, g.nnz
    ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
    for dt in (torch.float32, torch.bfloat16):
        feat = torch.randn(n, h, f, device=DEV).to(dt)
        s = 4 if dt == torch.float32 else 2
        nbytes = nnz * (4 + h * 4 + h * f * s) + n * (4 + 2 * h * 4 + h * f * s)
        ref = None
        for vec in (0, 1, 2, 4, 8):
            for fe in (0, 1):
                _lib.hip().cogdl_hip_set_tuning(4, vec)
                _lib.hip().cogdl_hip_set_tuning(5, fe)
                try:
                    ms = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), 10)
                except Exception as e:  # workspace too small for a forced geometry etc.
                    print(tag, dt, "vec", vec, "fast_exp", fe, "ERR", e)
                    continue
                out = gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat)[0].float()
                if ref is None:
                    ref = out
                err = ((out - ref).abs().max() / ref.abs().max()).item()
                print("%s H=%d F=%d %s vec=%d fast_exp=%d: %8.1f us  %5.1f%% of 8TB/s  maxdiff_vs_auto %.1e" % (
                    tag, h, f, str(dt)[6:], vec, fe, ms * 1e3, nbytes / ms / 1e6 / 80, err), flush=True)
    _lib.hip().cogdl_hip_set_tuning(4, 0)
    _lib.hip().cogdl_hip_set_tuning(5, 0)


for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
    run(g, "arxiv-" + topo, 8, 8)
    run(g, "arxiv-" + topo, 4, 32)
n = 232_965
src, dst = synth.rmat_pairs(n, 57_300_000, seed=0, device=DEV)
g = synth.finalize(src, dst, n, norm=None)
del src, dst
run(g, "reddit-rmat", 8, 8)
