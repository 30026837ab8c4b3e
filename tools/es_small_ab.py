#!/usr/bin/env python3
"""A/B of the flat edge_softmax kernel's tile size on SMALL problems (arxiv-sized graphs): quarter-size tiles (default
below 4096 full tiles' worth of elements) against the full-size tiles (tuning key 9 bit 6).  HIP-event time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

lib = _lib.hip()
for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to("cuda:0")
    for h in (1, 2, 8, 16):
        for dt in (torch.float32, torch.bfloat16):
            a = torch.randn(g.nnz, h, device="cuda:0").to(dt)
            line = "arxiv-%-7s H=%-2d %-8s" % (topo, h, str(dt)[6:])
            outs = []
            for bit, label in ((0, "quarter tiles"), (64, "full tiles")):
                lib.cogdl_hip_set_tuning(9, bit)
                sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
                gin = es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, a)
                outs.append((sm, gin))
                f = timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a), 30)
                b = timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, a), 30)
                line += "  %s: fwd %6.1f us bwd %6.1f us" % (label, f * 1e3, b * 1e3)
            lib.cogdl_hip_set_tuning(9, 0)
            same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            print(line + ("  (identical results)" if same else "  (results differ in the last bits: other piece boundaries)"), flush=True)
