#!/usr/bin/env python3
"""Where does a one-row tile of the flat edge_softmax forward kernel spend its time?  Tuning key 9 bit 7 makes thread 0 of
every 61st one-row tile add the 10 ns ticks between five marks to the workspace header (waits forced at the marks)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators import edge_softmax as es_mod  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
g = synth.reddit_like(seed=0, device=DEV)
names = ("loads landed", "max pass + reduce", "exp pass + reduce", "publish / wait / merge", "scale + stores")
for h in (8, 1):
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(g.nnz, h, device=DEV).to(dt)
        for dbg, label in ((128, "default"), (129, "no exchange")):
            lib.cogdl_hip_set_tuning(9, dbg)
            es_mod._launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
            es_mod._launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
            torch.cuda.synchronize()
            hdr = es_mod.LAST_WORKSPACE[:64].view(torch.int32).cpu().tolist()
            n = max(hdr[15], 1)
            ms = timeit(lambda: es_mod._launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a), 5)
            parts = ["%s %.2f" % (nm, (hdr[8 + k] & 0xFFFFFFFF) / n / 100.0) for k, nm in enumerate(names)]
            tot = sum((hdr[8 + k] & 0xFFFFFFFF) for k in range(5)) / n / 100.0
            print("reddit H=%d %-8s %-11s %7.1f us/launch | one-row tiles sampled %d, us per tile: %s | sum %.2f" % (
                h, str(dt)[6:], label, ms * 1e3, hdr[15], ", ".join(parts), tot), flush=True)
        lib.cogdl_hip_set_tuning(9, 0)
