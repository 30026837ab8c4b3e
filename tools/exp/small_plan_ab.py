"""csr_spmm on arxiv-sized graphs: the ordinary launch (row blocks + long-row workgroups) against virtual rows in order of
length (an XCD plan, built at several split / piece settings).  Is the plan layout the better schedule for skewed graphs even
where the table is not cache-sized?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.spmm import csr_spmm_raw, csr_spmm_xcd_raw
from tools.ops_bench import timeit_graph
dev = "cuda:0"
for name, g in (("arxiv-uniform", synth.arxiv_like(0)), ("arxiv-rmat", synth.arxiv_like(0, "rmat"))):
    rp, ci, w32 = g.rowptr.to(dev), g.colind.to(dev), g.weight.to(dev)
    plans = {}
    for split, piece in ((128, 128), (128, 256), (64, 64), (32, 128), (1 << 20, 1 << 20)):
        plans[(split, piece)] = xcdplan.build(rp, ci, split=split, piece=piece)
    for dt in (torch.float32, torch.bfloat16):
        for f in (128, 64, 40):
            x = torch.randn(g.num_nodes, f, device=dev).to(dt); w = w32.to(dt)
            ref = csr_spmm_raw(rp, ci, w, x)
            out = ["plain %7.1f" % (timeit_graph(lambda: csr_spmm_raw(rp, ci, w, x)) * 1e3)]
            for key, plan in plans.items():
                wp = plan.permuted_values(w)
                got = csr_spmm_xcd_raw(plan, w, x)
                err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
                out.append("%s (%d parts) %7.1f (err %.0e)" % ("/".join(str(k) for k in key), plan.n_parts, timeit_graph(lambda: csr_spmm_xcd_raw(plan, w, x)) * 1e3, err))
            print("%-14s %-8s F=%-4d %s" % (name, str(dt)[6:], f, "   ".join(out)), flush=True)
