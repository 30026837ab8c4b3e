import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import _lib, synth
from cogdl_amd.operators.spmm import csr_spmm_raw
from tools.ops_bench import timeit_graph
dev = "cuda:0"; lib = _lib.hip()
for topo in ("uniform", "rmat"):
    g = synth.arxiv_like(seed=0, topology=topo).to(dev)
    n = g.num_nodes
    for dt in (torch.float32, torch.bfloat16):
        for f in (20, 24, 40, 48, 80, 64):
            x = torch.randn(n, f, device=dev).to(dt); w = g.weight.to(dt)
            res = []
            for key in (0, -99):
                lib.cogdl_hip_set_tuning(6, key)
                res.append(timeit_graph(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x)) * 1e3)
            lib.cogdl_hip_set_tuning(6, 0)
            b = g.nnz * (8 + f * x.element_size()) + n * (4 + f * x.element_size())
            print("%-8s %-9s F=%-3d  narrow groups %7.1f us (%.2f)   power-of-two groups %7.1f us (%.2f)" % (topo, str(dt)[6:], f, res[0], b / res[0] / 8e6, res[1], b / res[1] / 8e6), flush=True)
