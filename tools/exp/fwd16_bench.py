#!/usr/bin/env python3
"""cogdl_hip_linear_fwd_bf16 beside torch's autocast product on the shapes of BASELINE configs[2] (Reddit-shaped GAT)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import linear as cl  # noqa: E402

dev = "cuda:0"


def ms(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, k, n, xdt in ((232965, 602, 64, torch.float32), (232965, 64, 41, torch.bfloat16), (169343, 128, 64, torch.float32), (2449029, 100, 47, torch.float32)):
    x = torch.randn(rows, k, device=dev).to(xdt)
    w = torch.randn(k, n, device=dev) * 0.05
    wb = w.bfloat16()
    g = torch.randn(rows, n, device=dev).bfloat16()
    t_cast = ms(lambda: x.bfloat16()) if xdt == torch.float32 else 0.0
    xb = x.bfloat16()
    t_mm = ms(lambda: torch.mm(xb, wb))
    t_ours = ms(lambda: cl.tall_skinny_matmul_bf16(x, w, None, False))
    t_wg_torch = ms(lambda: torch.mm(xb.t(), g))
    xf, gf = x.float(), g.float()
    t_wg = ms(lambda: cl.linear_wgrad(xf, gf, want_bias=False))
    t_gcast = ms(lambda: g.float())
    byts = rows * k * x.element_size() + rows * n * 2
    print("%8d x %4d -> %2d  x %-8s  forward: torch cast %.0f + mm %.0f us   ours %.0f us (%.2f TB/s)   |  grad_W: torch bf16 mm %.0f us   fp32 MFMA wgrad %.0f us (+ %.0f us cast of grad)"
          % (rows, k, n, str(xdt).split(".")[1], t_cast, t_mm, t_ours, byts / t_ours / 1e6, t_wg_torch, t_wg, t_gcast))
