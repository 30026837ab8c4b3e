import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import _lib, synth
from cogdl_amd.operators.spmm import csr_spmm_raw
from tools.ops_bench import timeit_graph
dev = "cuda:0"; lib = _lib.hip()
g = synth.arxiv_like(seed=0, topology="rmat").to(dev)
n = g.num_nodes
for f in (40, 64, 128):
    x = torch.randn(n, f, device=dev); w = g.weight
    lib.cogdl_hip_set_tuning(6, -99 if f == 40 else 0)
    fn = lambda: csr_spmm_raw(g.rowptr, g.colind, w, x)
    lib.cogdl_hip_set_tuning(13, 1); t_long = timeit_graph(fn) * 1e3
    lib.cogdl_hip_set_tuning(13, 2); t_rows = timeit_graph(fn) * 1e3
    lib.cogdl_hip_set_tuning(13, 0)
    print("F=%d default %.1f us   long-row workgroups alone %.1f   row blocks alone %.1f" % (f, timeit_graph(fn) * 1e3, t_long, t_rows), flush=True)
    for thr in (64, 128, 256, 512):
        out = []
        for grid in (512, 1024, 1536, 2040):
            lib.cogdl_hip_set_tuning(1, thr); lib.cogdl_hip_set_tuning(3, grid)
            out.append("%7.1f" % (timeit_graph(fn) * 1e3))
        print("   thresh %-4d  us at long grid 512/1024/1536/2040: %s" % (thr, " ".join(out)), flush=True)
    lib.cogdl_hip_set_tuning(1, 0); lib.cogdl_hip_set_tuning(3, 1024)
