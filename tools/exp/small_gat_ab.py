"""Fused GAT forward + backward on the arxiv-sized graphs: ordinary kernels (COGDL_AMD_XCD=off) against XCD plans (force)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func
from tools.ops_bench import timeit
dev = "cuda:0"
for name, g in (("arxiv-uniform", synth.arxiv_like(0)), ("arxiv-rmat", synth.arxiv_like(0, "rmat"))):
    rp, ci = g.rowptr.to(dev), g.colind.to(dev); n = g.num_nodes
    for h, f, dt in ((8, 8, torch.bfloat16), (8, 8, torch.float32), (1, 40, torch.bfloat16), (1, 40, torch.float32), (4, 32, torch.float32)):
        ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
        ft = torch.randn(n, h, f, device=dev).to(dt).requires_grad_(); grad = torch.randn(n, h, f, device=dev).to(dt)
        for p in (0.0, 0.5):
            out = []
            for mode in ("off", "force"):
                xcdplan.MODE = mode
                fw = lambda: fused_gat_dropout_func(ar.detach(), ac.detach(), rp, ci, 0.2, ft.detach(), p, seed=3)
                def step():
                    torch.autograd.grad(fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, p, seed=3), (ar, ac, ft), grad)
                fw(); step(); torch.cuda.synchronize()
                out.append("%s: fwd %6.0f  fwd+bwd %6.0f us" % (mode, timeit(fw, 20) * 1e3, timeit(step, 20) * 1e3))
            print("%-14s gat H=%d F=%-3d %-8s p=%.1f   %s" % (name, h, f, str(dt)[6:], p, "   ".join(out)), flush=True)
