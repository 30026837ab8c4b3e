import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.spmm import csr_spmm_raw, csr_spmm_xcd_raw
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
plans = {}
for split, piece in ((64, 256), (256, 256), (1024, 256), (1024, 512), (4096, 512)):
    plans[(split, piece)] = xcdplan.build(g.rowptr, g.colind, split=split, piece=piece)
for dt in (torch.bfloat16, torch.float32):
    for f in (64, 128):
        x = torch.randn(n, f, device=dev).to(dt); w = g.weight.to(dt)
        t0 = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10) * 1e3
        out = ["plain %8.1f" % t0]
        for key, plan in plans.items():
            wp = plan.permuted_values(w)
            t = timeit(lambda: csr_spmm_xcd_raw(plan, w, x), 10) * 1e3
            out.append("split %d piece %d (%d parts) %8.1f" % (key[0], key[1], plan.n_parts, t))
        print("csr_spmm %s F=%d: " % (str(dt)[6:], f) + "   ".join(out), flush=True)
