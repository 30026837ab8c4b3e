#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over the operators whose ALGORITHMIC-byte fraction exceeds 100 % of the HBM
peak (SURVEY.md section 8d prices one gathered row per edge; registers, L2 and the Infinity Cache serve part of it): the
1 GiB calibration copy, then -- on the Reddit-shaped graph at its true size -- csr_spmm F=64, mhspmm / fused GAT
forward / mhsddmm / fused GAT backward at H=8 x F=8, csr_sddmm F=64; three launches each.  tools/pmc_by_kernel.py folds
the counters per kernel: the HBM-side bytes per launch and their rate are the fractions DESIGN.md quotes next to the
algorithmic ones.

ROUND 4: the fault of round 3 was not this library's (it fell 0.34 s after HSA initialisation, before the
first operator could run; the combined probe runs clean on other boxes) -- `gpu_round.sh pmccombined` runs everything in
one process again.  Two operators were added: gat_bf16 / gat_drop_bf16 (forward + backward through the autograd operator).

ROUND 3 STATUS: run with ALL operators in one process under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` this probe ended
with "Memory access fault by GPU node" and hung until the time limit (30 GPU-minutes lost); run ONE operator per process
(`python tools/pmc_probe_ops.py <name>`, under a 75 s `timeout`) every operator profiles cleanly -- that is how
profiles/r03_pmc_ops.json was collected.  The fault of the combined run is not diagnosed.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from cogdl_amd.operators.fused_gat import FusedGATFunction, fused_gat_dropout_func, gat_forward  # noqa: E402
from cogdl_amd.operators.mhspmm import mhsddmm_raw, mhspmm_raw  # noqa: E402
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw  # noqa: E402

dev = "cuda:0"
a = torch.randn(256 * 1024 * 1024, device=dev)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
del a, b
g = synth.reddit_like(seed=0, device=dev)
n, nnz, h, f = g.num_nodes, g.nnz, 8, 8
x64 = torch.randn(n, 64, device=dev)
y64 = torch.randn(n, 64, device=dev)
att = torch.randn(nnz, h, device=dev)
sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, att)
feat = torch.randn(n, h, f, device=dev)
grad = torch.randn(n, h, f, device=dev)
ar, ac = torch.randn(n, h, device=dev), torch.randn(n, h, device=dev)
ar_g, ac_g, ft_g = ar.clone().requires_grad_(), ac.clone().requires_grad_(), feat.clone().requires_grad_()
OPS = {
    "csr_spmm": lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x64),
    "mhspmm": lambda: mhspmm_raw(g.rowptr, g.colind, sm, feat),
    "gat_fwd": lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat),
    "mhsddmm": lambda: mhsddmm_raw(g.rowptr, g.colind, grad, feat),
    "csr_sddmm": lambda: csr_sddmm_raw(g.rowptr, g.colind, y64, x64),
    "gat_bwd": lambda: torch.autograd.grad(FusedGATFunction.apply(ar_g, ac_g, g.rowptr, g.colind, g.rowptr, g.colind, 0.2, ft_g),
                                           (ar_g, ac_g, ft_g), grad),
}
# round 4: configs[2]'s own shapes in bf16 through the autograd operator, without and with the attention dropout
ftb = feat.bfloat16()
gradb = grad.bfloat16()
ar_b, ac_b, ft_b = ar.clone().requires_grad_(), ac.clone().requires_grad_(), ftb.clone().requires_grad_()
for p_drop, tag in ((0.0, "gat_bf16"), (0.5, "gat_drop_bf16")):
    OPS[tag] = (lambda p_drop=p_drop: torch.autograd.grad(
        fused_gat_dropout_func(ar_b, ac_b, g.rowptr, g.colind, 0.2, ft_b, p_drop, seed=3), (ar_b, ac_b, ft_b), gradb))
which = sys.argv[1:] or list(OPS)  # (one operator per process narrows a fault under the profiler down)
for name in which:
    for _ in range(3):
        OPS[name]()
    torch.cuda.synchronize()
    print("done", name, flush=True)
print("reddit nnz", nnz)
