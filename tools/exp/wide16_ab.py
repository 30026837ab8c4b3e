"""A/B inside one process: 16-bit csr_spmm over the XCD-partitioned plan with 8-byte lanes (tuning key 6 = 4: the geometry of
before) against 16-byte lanes (default now), Reddit-shaped graph, default plan parameters; correctness against fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import _lib, synth, xcdplan
from cogdl_amd.operators.spmm import csr_spmm_raw, csr_spmm_xcd_raw
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
plan = xcdplan.build(g.rowptr, g.colind)
lib = _lib.hip()
for dt in (torch.bfloat16, torch.float16):
    for f in (32, 64, 128, 256, 40, 602):
        x = torch.randn(n, f, device=dev).to(dt); w = g.weight.to(dt)
        ref = csr_spmm_raw(g.rowptr, g.colind, g.weight, x.float())
        res = []
        for name, key in (("8-byte lanes", 4), ("16-byte lanes", 0), ("8-byte lanes", 4), ("16-byte lanes", 0)):
            lib.cogdl_hip_set_tuning(6, key)
            out = csr_spmm_xcd_raw(plan, w, x)
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            t = timeit(lambda: csr_spmm_xcd_raw(plan, w, x), 10) * 1e3
            res.append("%s %8.1f us (err %.1e)" % (name, t, err))
        lib.cogdl_hip_set_tuning(6, 0)
        print("csr_spmm_xcd %-8s F=%-4d %s" % (str(dt)[6:], f, "   ".join(res)), flush=True)
