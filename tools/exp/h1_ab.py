import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import _lib, synth
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func
from tools.ops_bench import timeit
dev = "cuda:0"; lib = _lib.hip()
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
for h, f in ((1, 41), (1, 64), (1, 16)):
    ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
    ft = torch.randn(n, h, f, device=dev).bfloat16().requires_grad_()
    grad = torch.randn(n, h, f, device=dev).bfloat16()
    for p in (0.0, 0.5):
        res = []
        for cap in (8, 0):
            lib.cogdl_hip_set_tuning(4, cap)
            fw = lambda: fused_gat_dropout_func(ar.detach(), ac.detach(), g.rowptr, g.colind, 0.2, ft.detach(), p, seed=3)
            t_f = timeit(fw, 10) * 1e3
            def step():
                o = fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, p, seed=3)
                torch.autograd.grad(o, (ar, ac, ft), grad)
            t_s = timeit(step, 10) * 1e3
            res.append((t_f, t_s - t_f))
        lib.cogdl_hip_set_tuning(4, 0)
        print("H=%d F=%d bf16 p=%.1f   8 lanes x 8: fwd %7.1f bwd %7.1f us    4 lanes x 16: fwd %7.1f bwd %7.1f us" % (h, f, p, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
