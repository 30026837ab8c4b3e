#!/usr/bin/env python3
"""Per-operator roofline table on the GPU box (not part of the product): every entry point of
include/cogdl_hip.h timed with HIP events on the workloads of SURVEY.md section 8 and priced with that
section's ALGORITHMIC bytes.  Usage:  python tools/ops_bench.py [arxiv] [reddit] [--json out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import _launch as es_launch  # noqa: E402
from cogdl_amd.operators.fused_gat import FusedGATFunction, gat_forward  # noqa: E402
from cogdl_amd.operators.mhspmm import mhsddmm_raw, mhspmm_raw  # noqa: E402
from cogdl_amd.operators.scatter_max import scatter_max_bp, scatter_max_bp_csc, scatter_max_fp  # noqa: E402
from cogdl_amd.operators.spmm import csr_sddmm_raw, csr_spmm_raw  # noqa: E402
from cogdl_amd.plan import csr2csc, gather_rows  # noqa: E402

DEV = "cuda:0"
PEAK = 8000.0
ROWS = []


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def timeit_graph(fn, inner=10, reps=7):
    """GPU time of one call, free of the host's launch cost: `inner` calls captured into one HIP graph, replayed."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(inner):
            fn()
    return timeit(graph.replay, reps, 2) / inner


def report(op, cfg, ms, nbytes, nnz):
    gbs = nbytes / ms / 1e6
    row = {"op": op, "config": cfg, "us": ms * 1e3, "alg_GB": nbytes / 1e9, "GBs": gbs, "frac": gbs / PEAK,
           "GEdges_s": nnz / ms / 1e6}
    ROWS.append(row)
    print("%-28s %-34s %10.1f us  %8.3f GB  %7.0f GB/s  %5.1f%%  %7.2f GEdges/s" % (
        op, cfg, row["us"], row["alg_GB"], gbs, 100 * row["frac"], row["GEdges_s"]), flush=True)


def arxiv():
    for topo in ("uniform", "rmat"):
        g = synth.arxiv_like(seed=0, topology=topo).to(DEV)
        n, nnz = g.num_nodes, g.nnz
        from cogdl_amd import _lib, xcdplan
        from cogdl_amd.operators.spmm import csr_spmm_xcd_raw
        # the launch a SKEWED structure takes once its fingerprint is known (xcdplan.ordered_wanted): virtual rows by length
        xplan = xcdplan.build(g.rowptr, g.colind, split=int(_lib.hip().cogdl_hip_exact_row_edges(nnz))) if topo == "rmat" else None
        for f in (40, 64, 128, 256):
            x = torch.randn(n, f, device=DEV)
            cfg = "arxiv-%s F=%d f32" % (topo, f)
            report("csr_spmm", cfg, timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x)),
                   nnz * (8 + f * 4) + n * (4 + f * 4), nnz)
            if xplan is not None:
                report("csr_spmm (plan: rows by length)", cfg, timeit(lambda: csr_spmm_xcd_raw(xplan, g.weight, x)),
                       nnz * (8 + f * 4) + n * (4 + f * 4), nnz)
                xb16, wb16 = x.bfloat16(), g.weight.bfloat16()
                report("csr_spmm (plan: rows by length)", "arxiv-%s F=%d bf16" % (topo, f),
                       timeit(lambda: csr_spmm_xcd_raw(xplan, wb16, xb16)), nnz * (6 + f * 2) + n * (4 + f * 2), nnz)
            if f in (64, 128):
                from cogdl_amd.operators.spmm import csr_spmm_epilogue_raw
                on, inn = torch.rand(n, 1, device=DEV) + 0.5, torch.rand(n, 1, device=DEV) + 0.5
                report("csr_spmm_epilogue(norms+relu)", cfg,
                       timeit(lambda: csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x, on, inn, None, True)),
                       nnz * (12 + f * 4) + n * (8 + f * 4), nnz)
                report("  unfused: out_norm*x, spmm, in_norm*., relu", cfg,
                       timeit(lambda: torch.relu(inn * csr_spmm_raw(g.rowptr, g.colind, g.weight, on * x))),
                       nnz * (12 + f * 4) + n * (8 + f * 4), nnz)
                report("csr_spmm(unweighted)", cfg, timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, None, x)),
                       nnz * (4 + f * 4) + n * (4 + f * 4), nnz)
                xb = x.bfloat16()
                wb = g.weight.bfloat16()
                report("csr_spmm", "arxiv-%s F=%d bf16" % (topo, f),
                       timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, wb, xb)),
                       nnz * (6 + f * 2) + n * (4 + f * 2), nnz)
                y = torch.randn(n, f, device=DEV)
                report("csr_sddmm", cfg, timeit(lambda: csr_sddmm_raw(g.rowptr, g.colind, y, x)),
                       nnz * (4 + 4 + 2 * f * 4 + 4), nnz)
                report("scatter_max_fwd", cfg, timeit(lambda: scatter_max_fp(g.rowptr, g.colind, x)),
                       nnz * (4 + f * 4) + n * (4 + f * 4 + f * 4), nnz)
                _, mid = scatter_max_fp(g.rowptr, g.colind, x)
                report("scatter_max_bwd(atomic)", cfg, timeit(lambda: scatter_max_bp(y, mid, n)), n * f * (4 + 4 + 4 + 4), nnz)
                tplan = csr2csc(g.rowptr, g.colind, n)
                report("scatter_max_bwd(gather)", cfg, timeit(lambda: scatter_max_bp_csc(tplan.colptr, tplan.rowind, y, mid, n)),
                       nnz * (4 + 2 * f * 4) + n * (4 + f * 4), nnz)
        # message operators (cogdl/operators/ops.py): fused kernel vs the reference's torch composition on the GPU
        from cogdl_amd.operators import ops as mops
        import types
        rows = torch.repeat_interleave(torch.arange(n, device=DEV), (g.rowptr[1:] - g.rowptr[:-1]).long())
        shuffle = torch.randperm(nnz, device=DEV)  # an UNSORTED edge list, as Graph(edge_index=...) holds it
        coo = types.SimpleNamespace(edge_index=(rows[shuffle].contiguous(), g.colind.long()[shuffle].contiguous()),
                                    edge_weight=g.weight[shuffle].contiguous())
        for f in (64, 128):
            x, ef = torch.randn(n, f, device=DEV), torch.randn(nnz, f, device=DEV)
            cfg = "arxiv-%s F=%d f32 COO" % (topo, f)
            mops.s_mul_e_sum(coo, x, ef, weight=True)  # builds and caches the destination plan
            if f > 64:  # (a memoised edge list of a skewed graph takes the length-ordered plan from its second use on, rows of
                from cogdl_amd import xcdplan  # more than 64 columns: the ordinary launch beside it)

                mode, xcdplan.MODE = xcdplan.MODE, "off"
                report("s_mul_e_sum(fused, ordinary launch)", cfg, timeit(lambda: mops.s_mul_e_sum(coo, x, ef, weight=True)),
                       nnz * (4 + 4 + 4 + 2 * f * 4) + n * (4 + f * 4), nnz)
                xcdplan.MODE = mode
            report("s_mul_e_sum(fused)", cfg, timeit(lambda: mops.s_mul_e_sum(coo, x, ef, weight=True)),
                   nnz * (4 + 4 + 4 + 2 * f * 4) + n * (4 + f * 4), nnz)

            def torch_composition():
                r, c = coo.edge_index
                msg = x[c] * ef * coo.edge_weight.view(-1, 1)
                return torch.zeros(n, f, device=DEV).scatter_add_(0, r.view(-1, 1).expand(nnz, f), msg)

            report("s_mul_e_sum(torch ref)", cfg, timeit(torch_composition),
                   nnz * (4 + 4 + 4 + 2 * f * 4) + n * (4 + f * 4), nnz)
            report("scatter_add(fused)", cfg, timeit(lambda: mops.scatter_add(ef, coo.edge_index[0], n)),
                   nnz * (4 + f * 4) + n * (4 + f * 4), nnz)
            report("scatter_add(torch ref)", cfg,
                   timeit(lambda: torch.zeros(n, f, device=DEV).scatter_add_(
                       0, coo.edge_index[0].view(-1, 1).expand(nnz, f), ef)),
                   nnz * (4 + f * 4) + n * (4 + f * 4), nnz)
        report("csr2csc", "arxiv-%s" % topo, timeit(lambda: csr2csc(g.rowptr, g.colind, n), reps=10),
               nnz * (4 + 4) * 2 + 8 * (n + 1), nnz)
        plan = csr2csc(g.rowptr, g.colind, n)
        report("gather_rows(weights)", "arxiv-%s" % topo, timeit(lambda: gather_rows(plan.perm, g.weight)),
               nnz * 12, nnz)
        for h in (1, 8):
            a = torch.randn(nnz, h, device=DEV)
            cfg = "arxiv-%s H=%d" % (topo, h)
            report("edge_softmax_fwd", cfg, timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)),
                   nnz * h * 8 + 4 * (n + 1), nnz)
            s = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)
            report("edge_softmax_bwd", cfg, timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, s, a)),
                   nnz * h * 12, nnz)
            # the same calls replayed from a HIP graph: GPU time without the Python wrapper's ~30 us per call
            report("edge_softmax_fwd(hipGraph)", cfg,
                   timeit_graph(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, a)), nnz * h * 8 + 4 * (n + 1), nnz)
            report("edge_softmax_bwd(hipGraph)", cfg,
                   timeit_graph(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, s, a)), nnz * h * 12, nnz)


def gat_suite(g, tag, h, f, dtypes=(torch.float32, torch.bfloat16), reps=10):
    n, nnz = g.num_nodes, g.nnz
    att = torch.randn(nnz, h, device=DEV)
    ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
    cfg0 = "%s H=%d F=%d" % (tag, h, f)
    report("edge_softmax_fwd", cfg0, timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, att), reps),
           nnz * h * 8 + 4 * (n + 1), nnz)
    sm = es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, att)
    report("edge_softmax_bwd", cfg0, timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, att), reps),
           nnz * h * 12, nnz)
    if torch.bfloat16 in dtypes:
        attb, smb = att.bfloat16(), sm.bfloat16()
        report("edge_softmax_fwd", cfg0 + " bf16", timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, attb), reps),
               nnz * h * 4 + 4 * (n + 1), nnz)
        report("edge_softmax_bwd", cfg0 + " bf16",
               timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, smb, attb), reps), nnz * h * 6, nnz)
    # A/B: the round-1 row kernels (tuning key 7, bit 2) on the same inputs
    from cogdl_amd import _lib
    _lib.hip().cogdl_hip_set_tuning(7, 4)
    report("edge_softmax_fwd(row kernels)", cfg0, timeit(lambda: es_launch("cogdl_hip_edge_softmax_fwd", g.rowptr, att), reps),
           nnz * h * 8 + 4 * (n + 1), nnz)
    report("edge_softmax_bwd(row kernels)", cfg0, timeit(lambda: es_launch("cogdl_hip_edge_softmax_bwd", g.rowptr, sm, att), reps),
           nnz * h * 12, nnz)
    _lib.hip().cogdl_hip_set_tuning(7, 0)
    for dt in dtypes:
        s = 4 if dt == torch.float32 else 2
        name = "f32" if dt == torch.float32 else "bf16"
        feat = torch.randn(n, h, f, device=DEV).to(dt)
        cfg = "%s %s" % (cfg0, name)
        report("mhspmm", cfg, timeit(lambda: mhspmm_raw(g.rowptr, g.colind, sm, feat), reps),
               nnz * (4 + h * 4 + h * f * s) + n * (4 + h * f * s), nnz)
        report("gat_fwd(fused)", cfg, timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), reps),
               nnz * (4 + h * 4 + h * f * s) + n * (4 + 2 * h * 4 + h * f * s), nnz)
    feat = torch.randn(n, h, f, device=DEV)
    grad = torch.randn(n, h, f, device=DEV)
    report("mhsddmm", cfg0 + " f32", timeit(lambda: mhsddmm_raw(g.rowptr, g.colind, grad, feat), reps),
           nnz * (4 + h * 4 + 2 * h * f * 4), nnz)
    report("csr2csc", tag, timeit(lambda: csr2csc(g.rowptr, g.colind, n), reps=3, warm=1), nnz * 16 + 8 * (n + 1), nnz)
    plan = csr2csc(g.rowptr, g.colind, n)
    report("gather_rows(att[E,H])", cfg0, timeit(lambda: gather_rows(plan.perm, sm), reps), nnz * (4 + 8 * h), nnz)
    # backward product of mhspmm: A^T with the attention read through the permutation inside the kernel
    report("mhspmm(A^T, att[perm] in-kernel)", cfg0 + " f32",
           timeit(lambda: mhspmm_raw(plan.colptr, plan.rowind, sm, grad, eid=plan.perm), reps),
           nnz * (4 + 4 + h * 4 + h * f * 4) + n * (4 + h * f * 4), nnz)
    # fused backward through the autograd function (plan cached after the first call)
    ar_g, ac_g, ft_g = ar.clone().requires_grad_(), ac.clone().requires_grad_(), feat.clone().requires_grad_()

    def fwd_bwd():
        out = FusedGATFunction.apply(ar_g, ac_g, g.rowptr, g.colind, g.rowptr, g.colind, 0.2, ft_g)
        torch.autograd.grad(out, (ar_g, ac_g, ft_g), grad)

    t_fb = timeit(fwd_bwd, reps)
    t_f = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, feat), reps)
    # backward: two passes over the edges (CSR pass for grad_attn_row, CSC pass for grad_feat/grad_attn_col)
    report("gat_bwd(fused, fwd+bwd - fwd)", cfg0 + " f32", t_fb - t_f,
           2 * nnz * (4 + 2 * h * f * 4 + h * 4) + 4 * n * h * f * 4, nnz)
    if torch.bfloat16 in dtypes:  # configs[2]'s dtype: feat / out / grad read and written as bf16
        ftb, gb = feat.bfloat16().requires_grad_(), grad.bfloat16()

        def fwd_bwd16():
            out = FusedGATFunction.apply(ar_g, ac_g, g.rowptr, g.colind, g.rowptr, g.colind, 0.2, ftb)
            torch.autograd.grad(out, (ar_g, ac_g, ftb), gb)

        t_fb = timeit(fwd_bwd16, reps)
        t_f = timeit(lambda: gat_forward(ar, ac, g.rowptr, g.colind, 0.2, ftb.detach()), reps)
        report("gat_bwd(fused, fwd+bwd - fwd)", cfg0 + " bf16", t_fb - t_f,
               2 * nnz * (4 + 2 * h * f * 2 + h * 4) + 4 * n * h * f * 2, nnz)


def reddit():
    """Reddit-shaped GAT workload (configs[2]): N=232,965, 114,848,857 nnz (R-MAT rows, Reddit's edge count), H=8 x F=8;
    generated on the GPU."""
    n = synth.REDDIT_NODES
    g = synth.reddit_like(seed=0, device=DEV)
    print("reddit-like: N=%d nnz=%d max_deg=%d" % (n, g.nnz, int(g.degrees().max())), flush=True)
    gat_suite(g, "reddit-rmat", 8, 8)
    gat_suite(g, "reddit-rmat", 1, 41, dtypes=(torch.float32,))
    x = torch.randn(n, 128, device=DEV)
    w = torch.rand(g.nnz, device=DEV)
    report("csr_spmm", "reddit-rmat F=128 f32", timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10),
           g.nnz * (8 + 512) + n * 516, g.nnz)
    xb, wb = x.bfloat16(), w.bfloat16()
    report("csr_spmm", "reddit-rmat F=128 bf16", timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, wb, xb), 10),
           g.nnz * (6 + 256) + n * 260, g.nnz)


def linear():
    """The dense side: weight/bias gradient on tall-skinny operands; bytes = K*(in+out)*4."""
    from cogdl_amd.linear import linear_wgrad

    for k, i, o in ((169_343, 128, 64), (169_343, 64, 40), (169_343, 256, 256), (232_965, 602, 64), (2_449_029, 100, 128)):
        x, g = torch.randn(k, i, device=DEV), torch.randn(k, o, device=DEV)
        ms = timeit(lambda: linear_wgrad(x, g), 20)
        ms_t = timeit(lambda: (g.t() @ x, g.sum(0)), 10)
        nbytes = k * (i + o) * 4
        report("linear_wgrad (MFMA)", "K=%d in=%d out=%d" % (k, i, o), ms, nbytes, k)
        print("    torch g.t() @ x + g.sum(0): %.1f us;  %.1f TFLOP/s fp32 MFMA (peak 157)" % (ms_t * 1e3, 2.0 * k * i * o / ms / 1e9),
              flush=True)
        from cogdl_amd.linear import tall_skinny_matmul

        w, b = torch.randn(o, i, device=DEV), torch.randn(o, device=DEV)
        if tall_skinny_matmul(x, w, b, True) is not None:
            ms = timeit(lambda: tall_skinny_matmul(x, w, b, True), 20)
            ms_t = timeit(lambda: torch.addmm(b, x, w.t()), 10)
            report("linear_fwd (MFMA)", "K=%d in=%d out=%d" % (k, i, o), ms, nbytes, k)
            print("    torch addmm: %.1f us;  %.1f TFLOP/s" % (ms_t * 1e3, 2.0 * k * i * o / ms / 1e9), flush=True)
            if tall_skinny_matmul(g, w, None, False) is not None:  # declined (-> hipBLASLt) beyond 64 output columns
                ms = timeit(lambda: tall_skinny_matmul(g, w, None, False), 20)
                ms_t = timeit(lambda: g @ w, 10)
                report("linear_dgrad (MFMA)", "K=%d in=%d out=%d" % (k, i, o), ms, nbytes, k)
                print("    torch g @ w: %.1f us" % (ms_t * 1e3), flush=True)


def linear_bf16():
    """The same product under bf16 autocast (csrc/linear_fwd16.hip): bytes = x as the model holds it + the bf16 result."""
    from cogdl_amd.linear import tall_skinny_matmul_bf16

    for k, i, o, xdt in ((232_965, 602, 64, torch.float32), (232_965, 64, 41, torch.bfloat16), (169_343, 128, 64, torch.float32),
                         (2_449_029, 100, 47, torch.float32)):
        x, w = torch.randn(k, i, device=DEV).to(xdt), torch.randn(i, o, device=DEV)
        ms = timeit(lambda: tall_skinny_matmul_bf16(x, w, None, False), 20)
        xb, wb = x.bfloat16(), w.bfloat16()
        ms_t = timeit(lambda: torch.mm(xb, wb), 10)
        ms_c = timeit(lambda: x.bfloat16(), 10) if xdt == torch.float32 else 0.0
        report("linear_fwd_bf16 (MFMA)", "K=%d in=%d out=%d x=%s" % (k, i, o, str(xdt).split(".")[1]), ms,
               k * i * x.element_size() + k * o * 2, k)
        print("    torch (autocast): cast of x %.1f us + bf16 mm %.1f us" % (ms_c * 1e3, ms_t * 1e3), flush=True)


def main():
    argv = sys.argv[1:]
    json_path = None
    if "--json" in argv:
        i = argv.index("--json")
        json_path = argv[i + 1]
        del argv[i:i + 2]
    args = argv
    print(torch.cuda.get_device_name(0))
    if not args or "arxiv" in args:
        arxiv()
        g = synth.arxiv_like(seed=0, topology="rmat").to(DEV)
        gat_suite(g, "arxiv-rmat", 8, 8)
        gat_suite(g, "arxiv-rmat", 4, 32, dtypes=(torch.float32,))
    if not args or "reddit" in args:
        reddit()
    if not args or "linear" in args:
        linear()
        linear_bf16()
    if json_path:
        json.dump(ROWS, open(json_path, "w"), indent=1)


if __name__ == "__main__":
    main()
