import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.spmm import SPMMFunction
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
for f in (32, 64, 128, 256):
    x = torch.randn(n, f, device=dev).requires_grad_(); gout = torch.randn(n, f, device=dev)
    res = []
    for mode in ("off", "auto"):
        xcdplan.MODE = mode
        fw = lambda: SPMMFunction.apply(g.rowptr, g.colind, x.detach(), g.weight, False)
        def step():
            torch.autograd.grad(SPMMFunction.apply(g.rowptr, g.colind, x, g.weight, False), x, gout)
        fw(); step(); torch.cuda.synchronize()
        res.append((timeit(fw, 10) * 1e3, timeit(step, 10) * 1e3))
    print("csr_spmm fp32 F=%-3d reddit-shaped   plan off: fwd %8.1f us  fwd+bwd %8.1f us    plan on (split = exact-row bound): fwd %8.1f us  fwd+bwd %8.1f us" % (f, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
