import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func
from tools.ops_bench import timeit
dev = "cuda:0"
g = synth.reddit_like(seed=0, device=dev, norm="sym"); n = g.num_nodes
for h, f, dt in ((8, 8, torch.bfloat16), (1, 41, torch.bfloat16), (8, 8, torch.float32)):
    ar, ac = torch.randn(n, h, device=dev).requires_grad_(), torch.randn(n, h, device=dev).requires_grad_()
    ft = torch.randn(n, h, f, device=dev).to(dt).requires_grad_(); grad = torch.randn(n, h, f, device=dev).to(dt)
    for p in (0.0, 0.5):
        out = []
        for split, piece in ((64, 128), (128, 128), (256, 128), (64, 256), (128, 256), (256, 256), (256, 512), (512, 512)):
            xcdplan.SPLIT, xcdplan.PIECE = split, piece
            fw = lambda: fused_gat_dropout_func(ar.detach(), ac.detach(), g.rowptr, g.colind, 0.2, ft.detach(), p, seed=3)
            def step():
                torch.autograd.grad(fused_gat_dropout_func(ar, ac, g.rowptr, g.colind, 0.2, ft, p, seed=3), (ar, ac, ft), grad)
            fw(); step(); torch.cuda.synchronize()
            out.append("%d/%d: %6.0f %6.0f" % (split, piece, timeit(fw, 10) * 1e3, timeit(step, 10) * 1e3))
        xcdplan.XPLANS.clear()
        print("gat H=%d F=%d %s p=%.1f  split/piece: fwd, fwd+bwd us   " % (h, f, str(dt)[6:], p) + "   ".join(out), flush=True)
