#!/usr/bin/env python3
"""Experiment driver: time the fused GAT forward (plan off / on) with a variant library.  usage: run_exp.py <lib.so|-> """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import _lib
if sys.argv[1] != "-":
    _lib.HIP_LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym")
n = g.num_nodes
for dt in (torch.bfloat16, torch.float32):
    h, f = 8, 8
    ar, ac = torch.randn(n, h, device=dev), torch.randn(n, h, device=dev)
    ft = torch.randn(n, h, f, device=dev).to(dt)
    res = []
    for mode in ("off", "auto"):
        xcdplan.MODE = mode
        fn = lambda: fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, 0.0, seed=3)
        fn(); torch.cuda.synchronize()
        res.append(timeit(fn, 10) * 1e3)
    print("%s gat fwd H8F8 %s   plan off %8.1f us   plan on %8.1f us" % (sys.argv[1], str(dt)[6:], res[0], res[1]), flush=True)
