#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (CogDL) on CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference package is imported from a scratch COPY (importing it in place writes __pycache__
and hipified sources into the source tree) with the four absent third-party modules stubbed
(tests/golden/_stubs: optuna, numba, grave, turtle).  Nothing from the reference is copied into
the repo -- only the numeric inputs/outputs of its public functions:

  docs_csr.npz        Graph(edge_index) -> CSR           docs/source/tutorial/graph.rst:53-61
  spmm_cpu.npz        cogdl.operators.spmm.spmm_cpu      operators/spmm/spmm_cpu.cpp (CogDL's own JIT build)
  spmm_scatter.npz    spmm_scatter fwd + autograd grads  utils/spmm_utils.py:43-52
  sampler.npz         coo2csr_cpu(_index), sample_adj(-1), subgraph      operators/sample/sample.cpp
  edge_softmax.npz    edge_softmax (CPU fallback)        utils/spmm_utils.py:149-188
  gcn_layer.npz       Graph.sym_norm + GCNLayer fwd/bwd  layers/gcn_layer.py:51-64
  gat_layer.npz       GATLayer fwd/bwd (unfused path)    layers/gat_layer.py:59-86
  sage_layer.npz      Graph.sample_adj(-1) + SAGELayer(mean) fwd   layers/sage_layer.py:8-12,69-87
  message_ops.npz     scatter_add, s_*_e_sum / s_*_e_mean (+ autograd grads) on an UNSORTED COO graph   operators/ops.py:4-103

`python tests/golden/make_golden.py NAME...` rewrites only the named fixtures (every fixture draws from the same seeded
generator in file order, so the others stay what they were).
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("COGDL_REFERENCE", "/root/reference")


def import_reference():
    scratch = tempfile.mkdtemp(prefix="cogdl_refcopy_")
    shutil.copytree(os.path.join(REF, "cogdl"), os.path.join(scratch, "cogdl"))
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(HERE, "_stubs"), scratch]
    import cogdl  # noqa: F401

    return scratch


def main():
    import_reference()
    import torch
    from cogdl.data import Graph
    from cogdl.layers import GATLayer, GCNLayer, SAGELayer
    from cogdl.operators.sample import coo2csr_cpu, coo2csr_cpu_index, sample_adj_c
    from cogdl.operators import sample as ref_sample_mod
    from cogdl.operators.spmm import spmm_cpu
    from cogdl.utils import spmm_utils

    assert spmm_cpu is not None and sample_adj_c is not None, "reference native CPU ops failed to build"
    torch.manual_seed(1234)
    torch.set_num_threads(4)

    only = set(sys.argv[1:])

    def save(name, **arrays):
        if only and name not in only:
            return
        out = {}
        for k, v in arrays.items():
            out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote %-18s %s" % (name, {k: tuple(v.shape) for k, v in out.items()}))

    # ---- docs golden CSR (contiguous row/col; the doc's .t() view trips the stride bug) --------
    edges = torch.tensor([[0, 1], [1, 3], [2, 1], [4, 2], [0, 3]]).t().contiguous()
    g = Graph(edge_index=(edges[0].clone(), edges[1].clone()))
    save("docs_csr", row=edges[0], col=edges[1], row_indptr=g.row_indptr, col_indices=g.col_indices,
         edge_weight=g.edge_weight)

    # ---- csr_spmm_cpu -------------------------------------------------------------------------
    cases = {}
    gen = torch.Generator().manual_seed(7)
    for idx, (m, n_src, k, deg) in enumerate([(37, 37, 16, 5), (64, 80, 128, 9), (5, 9, 1, 3), (33, 33, 47, 4),
                                             (130, 100, 256, 70), (20, 20, 8, 0)]):
        d = torch.randint(0, 2 * deg + 1, (m,), generator=gen)
        d[torch.rand(m, generator=gen) < 0.15] = 0
        rowptr = torch.zeros(m + 1, dtype=torch.int32)
        rowptr[1:] = torch.cumsum(d, 0)
        nnz = int(rowptr[-1])
        colind = torch.randint(0, n_src, (nnz,), generator=gen, dtype=torch.int32)
        val = torch.randn(nnz, generator=gen)
        x = torch.randn(n_src, k, generator=gen)
        out = spmm_cpu(rowptr, colind, val, x)
        for nm, t in (("rowptr", rowptr), ("colind", colind), ("val", val), ("x", x), ("out", out)):
            cases["c%d_%s" % (idx, nm)] = t
    save("spmm_cpu", **cases)

    # ---- spmm_scatter forward + autograd gradients (the reference's CPU training path) --------
    n, k = 60, 24
    row = torch.randint(0, n, (400,), generator=gen)
    col = torch.randint(0, n, (400,), generator=gen)
    order = torch.argsort(row * n + col)  # COO sorted by (row, col) like a coalesced CogDL graph
    row, col = row[order], col[order]
    w = torch.randn(400, generator=gen).requires_grad_()
    b = torch.randn(n, k, generator=gen).requires_grad_()
    gout = torch.randn(n, k, generator=gen)
    out = spmm_utils.spmm_scatter(row, col, w, b)
    out.backward(gout)
    save("spmm_scatter", row=row, col=col, w=w, b=b, gout=gout, out=out, grad_b=b.grad, grad_w=w.grad)

    # ---- sampler module -----------------------------------------------------------------------
    n = 200
    row = torch.randint(0, n, (1500,), generator=gen)
    col = torch.randint(0, n, (1500,), generator=gen)
    val = torch.randn(1500, generator=gen)
    rp, ci, ov = coo2csr_cpu(row, col, val, n)
    rp2, perm = coo2csr_cpu_index(row, col, n)
    seeds = torch.randperm(n, generator=gen)[:32]
    s_indptr, s_indices, s_nodes, s_edges = sample_adj_c(rp, ci, seeds, -1, False)
    sub = torch.randperm(n, generator=gen)[:50]
    g_indptr, g_indices, g_nodes, g_edges = ref_sample_mod.sample.subgraph(rp, ci, sub)
    save("sampler", row=row, col=col, val=val, n=n, row_ptr=rp, col_ind=ci, out_val=ov, row_ptr_index=rp2, perm=perm,
         seeds=seeds, s_indptr=s_indptr, s_indices=s_indices, s_nodes=s_nodes, s_edges=s_edges, sub=sub,
         g_indptr=g_indptr, g_indices=g_indices, g_nodes=g_nodes, g_edges=g_edges)

    # ---- a small CogDL graph for the layer fixtures ---------------------------------------------
    n = 400
    src = torch.randint(0, n, (1100,), generator=gen)
    dst = torch.randint(0, n, (1100,), generator=gen)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = torch.unique(torch.cat([src * n + dst, dst * n + src]))  # symmetrise + coalesce (datasets/ogb.py:50-55)
    row, col = key // n, key % n

    def fresh_graph():
        gr = Graph(edge_index=(row.clone(), col.clone()), num_nodes=n)
        gr.add_remaining_self_loops()
        return gr

    # ---- edge_softmax CPU fallback (true softmax while max <= 10) -------------------------------
    gr = fresh_graph()
    ev = torch.randn(gr.num_edges, 3, generator=gen) * 2.0
    sm = spmm_utils.edge_softmax(gr, ev.clone())
    save("edge_softmax", row_indptr=gr.row_indptr, col_indices=gr.col_indices, values=ev, softmax=sm)

    # ---- GCNLayer: sym_norm + Linear + spmm ----------------------------------------------------
    gr = fresh_graph()
    gr.sym_norm()
    layer = GCNLayer(32, 16)
    x = torch.randn(n, 32, generator=gen)
    G = torch.randn(n, 16, generator=gen)
    with torch.no_grad():
        out_infer = layer(gr, x)  # x.requires_grad False -> csr_spmm_cpu path (spmm_utils.py:110-119)
    xg = x.clone().requires_grad_()
    out_train = layer(gr, xg)  # -> spmm_scatter path (spmm_utils.py:120-122)
    (out_train * G).sum().backward()
    save("gcn_layer", row_indptr=gr.row_indptr, col_indices=gr.col_indices, edge_weight=gr.edge_weight,
         x=x, G=G, W=layer.linear.weight, b=layer.linear.bias, out_infer=out_infer, out_train=out_train,
         grad_x=xg.grad, grad_W=layer.linear.weight.grad, grad_b=layer.linear.bias.grad)

    # ---- GATLayer (unfused path; attn_drop = 0 so it is deterministic) -------------------------
    gr = fresh_graph()
    gat = GATLayer(16, 8, nhead=4, attn_drop=0.0, alpha=0.2)
    x = torch.randn(n, 16, generator=gen)
    G = torch.randn(n, 32, generator=gen)
    xg = x.clone().requires_grad_()
    out = gat(gr, xg)
    (out * G).sum().backward()
    save("gat_layer", row_indptr=gr.row_indptr, col_indices=gr.col_indices, x=x, G=G, W=gat.W, a_l=gat.a_l,
         a_r=gat.a_r, out=out, grad_x=xg.grad, grad_W=gat.W.grad, grad_a_l=gat.a_l.grad, grad_a_r=gat.a_r.grad)

    # ---- GraphSAGE block: Graph.sample_adj(-1) + SAGELayer(mean) -------------------------------
    gr = fresh_graph()
    batch = torch.randperm(n, generator=gen)[:24]
    nodes, block = gr.sample_adj(batch, size=-1, replace=False)
    sage = SAGELayer(16, 12, aggr="mean")
    x_src = torch.randn(nodes.shape[0], 16, generator=gen)
    with torch.no_grad():
        out = sage(block, x_src)
    save("sage_layer", g_row_indptr=gr.row_indptr, g_col_indices=gr.col_indices, batch=batch, nodes=nodes,
         block_row_indptr=block.row_indptr, block_col_indices=block.col_indices, x_src=x_src,
         fc_W=sage.fc.weight, fc_b=sage.fc.bias, out=out)

    # ---- message operators (pure torch in the reference): unsorted COO, empty destinations, repeated edges ------
    from cogdl.operators import ops as ref_ops

    gen2 = torch.Generator().manual_seed(4321)
    n, e, f = 300, 2000, 12
    row = torch.randint(0, n - 20, (e,), generator=gen2)  # the last 20 nodes receive nothing
    col = torch.randint(0, n, (e,), generator=gen2)
    x = torch.randn(n, f, generator=gen2)
    ef = torch.randn(e, f, generator=gen2)
    es = torch.randn(e, generator=gen2)
    w = torch.rand(e, generator=gen2) + 0.5
    G = torch.randn(n, f, generator=gen2)
    gr = Graph(edge_index=(row.clone(), col.clone()), edge_weight=w.clone(), num_nodes=n)
    assert gr._adj.row_ptr is None  # still COO: the operators below never build the CSR
    res = {}
    for op1 in ("add", "sub", "mul"):
        for op2 in ("sum", "mean"):
            fn = getattr(ref_ops, "s_%s_e_%s" % (op1, op2))
            res["%s_%s" % (op1, op2)] = fn(gr, x, ef)
            res["%s_%s_w" % (op1, op2)] = fn(gr, x, ef, weight=True)
    res["mul_sum_scalar"] = ref_ops.s_mul_e_sum(gr, x, es)  # 1-D e_feat is viewed [E, 1] (ops.py:45-46)
    res["scatter_add"] = ref_ops.scatter_add(ef, row, n)
    xg, eg = x.clone().requires_grad_(), ef.clone().requires_grad_()
    (ref_ops.s_mul_e_mean(gr, xg, eg, weight=True) * G).sum().backward()
    res["grad_x_mul_mean_w"], res["grad_e_mul_mean_w"] = xg.grad, eg.grad
    xg, sg = x.clone().requires_grad_(), es.clone().requires_grad_()
    (ref_ops.s_sub_e_sum(gr, xg, sg) * G).sum().backward()
    res["grad_x_sub_sum_scalar"], res["grad_e_sub_sum_scalar"] = xg.grad, sg.grad
    assert gr._adj.row_ptr is None
    save("message_ops", row=row, col=col, n=n, x=x, ef=ef, es=es, w=w, G=G, **res)


if __name__ == "__main__":
    main()
