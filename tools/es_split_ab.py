#!/usr/bin/env python3
"""edge_softmax forward: the one-kernel form (one-row tiles hold their values across the cross-tile exchange) against the
round-5 two-kernel form (es_stream_kernel: two-pass streaming, 7-8 workgroups per CU) -- tuning key 9 bit 8 (opt-in) --
on the Reddit-shaped graph and the arxiv-sized R-MAT graph; kernel time by HIP events (graph replay of 10 calls: free of
the host's launch cost), algorithmic fraction of 8 TB/s (2 * E * H * s bytes), and equality of the two results."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch  # noqa: E402
from tools.ops_bench import timeit_graph  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()


def run(name, rowptr, nnz, heads, dtypes, modes):
    for h in heads:
        for dt in dtypes:
            v = (torch.randn(nnz, h, device=DEV) * 2).to(dt)
            res, outs = [], []
            for label, key in modes:
                lib.cogdl_hip_set_tuning(9, key)
                outs.append(_launch("cogdl_hip_edge_softmax_fwd", rowptr, v))
                ms = timeit_graph(lambda: _launch("cogdl_hip_edge_softmax_fwd", rowptr, v))
                res.append("%s %8.1f us (%4.1f %%)" % (label, ms * 1e3, 2 * nnz * h * v.element_size() / (ms * 1e-3) / 8e12 * 100))
            lib.cogdl_hip_set_tuning(9, 0)
            same = all(torch.allclose(o.float(), outs[0].float(), rtol=1e-2 if dt != torch.float32 else 1e-5, atol=1e-9) for o in outs)
            print("%-12s H=%-2d %-8s  %s   results agree: %s" % (name, h, str(dt)[6:], "   ".join(res), same), flush=True)
            del v, outs


g = synth.reddit_like(seed=0, device=DEV)
run("reddit-like", g.rowptr, g.nnz, (8, 1), (torch.float32, torch.bfloat16), (("default (one kernel)", 0), ("split", 256)))
del g
g = synth.arxiv_like(seed=0, topology="rmat").to(DEV)
run("arxiv-rmat", g.rowptr, g.nnz, (8, 1), (torch.float32, torch.bfloat16), (("default", 0), ("full tiles", 64), ("split+full tiles", 64 | 256)))
