"""Pin the oracle (oracle/cogdl_oracle.c) before anything is compared against it:
  * against the golden vectors produced by running the reference itself (tests/golden/*.npz),
  * against the reference's own C++ compiled in place (oracle/_ref), when present.
CPU only."""
import numpy as np
import pytest
import torch

from cogdl_amd import synth


def _cases(z):
    idx = sorted({k.split("_")[0] for k in z})
    return [{n: z["%s_%s" % (c, n)] for n in ("rowptr", "colind", "val", "x", "out")} for c in idx]


def test_docs_golden_csr(golden, oracle):
    # docs/source/tutorial/graph.rst:53-61
    z = golden("docs_csr")
    assert z["row_indptr"].tolist() == [0, 2, 3, 4, 4, 5]
    assert z["col_indices"].tolist() == [1, 3, 3, 1, 2]
    rp, perm = oracle.coo2csr_index(z["row"], z["col"], 5)
    assert rp.tolist() == z["row_indptr"].tolist()
    assert z["col"][perm].tolist() == z["col_indices"].tolist()


def test_spmm_oracle_bit_exact_vs_reference_goldens(golden, oracle):
    for c in _cases(golden("spmm_cpu")):
        out = oracle.csr_spmm(c["rowptr"], c["colind"], c["val"], c["x"])
        assert out.tobytes() == c["out"].tobytes()  # bit-exact
        out_mt = oracle.csr_spmm(c["rowptr"], c["colind"], c["val"], c["x"], nthreads=4)
        assert out_mt.tobytes() == c["out"].tobytes()


def test_spmm_oracle_vs_ref_build_arxiv_slice(oracle):
    if not oracle.ref_available("asshipped"):
        pytest.skip("oracle/_ref not built")
    ref = oracle.ref_spmm_cpu("asshipped")
    g = synth.scaled(5000, 12, seed=3)
    x = torch.randn(g.num_nodes, 128, generator=torch.Generator().manual_seed(0))
    want = ref(g.rowptr, g.colind, g.weight, x).numpy()
    got = oracle.csr_spmm(g.rowptr, g.colind, g.weight, x, nthreads=2)
    assert got.tobytes() == want.tobytes()
    # the -O3 -mavx2 -mfma build contracts to FMA: close, but NOT bitwise (documented in DESIGN.md)
    if oracle.ref_available("O3"):
        fast = oracle.ref_spmm_cpu("O3")(g.rowptr, g.colind, g.weight, x).numpy()
        np.testing.assert_allclose(fast, want, rtol=1e-5, atol=1e-6)


def test_spmm_scatter_golden_forward_and_gradients(golden, oracle):
    """The reference's CPU *training* path (spmm_scatter + autograd) vs the oracle's csr_spmm on A and on the
    stable transpose A^T: forward and grad_b are bit-exact, grad_w within 1e-5 (sddmm order unspecified)."""
    z = golden("spmm_scatter")
    n = z["b"].shape[0]
    rp, perm = oracle.coo2csr_index(z["row"], z["col"], n)
    rowptr, colind, w = rp.astype(np.int32), z["col"][perm].astype(np.int32), z["w"][perm]
    out = oracle.csr_spmm(rowptr, colind, w, z["b"])
    assert out.tobytes() == z["out"].tobytes()
    colptr, rowind, w_t, p = oracle.csr2csc(rowptr, colind, w, n_cols=n)
    grad_b = oracle.csr_spmm(colptr, rowind, w_t, z["gout"])
    assert grad_b.tobytes() == z["grad_b"].tobytes()
    grad_w = oracle.csr_sddmm(rowptr, colind, z["gout"], z["b"])
    np.testing.assert_allclose(grad_w, z["grad_w"][perm], rtol=1e-5, atol=1e-6)


def test_csr2csc_is_stable_and_consistent(oracle):
    g = synth.random_csr(40, 25, 6, seed=5)
    colptr, rowind, val_t, perm = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=25)
    dense = np.zeros((40, 25), np.float64)
    rows = np.repeat(np.arange(40), np.diff(g.rowptr.numpy()))
    np.add.at(dense, (rows, g.colind.numpy()), g.weight.numpy().astype(np.float64))
    dense_t = np.zeros((25, 40), np.float64)
    cols = np.repeat(np.arange(25), np.diff(colptr))
    np.add.at(dense_t, (cols, rowind), val_t.astype(np.float64))
    np.testing.assert_array_equal(dense.T, dense_t)
    for c in range(25):  # stability: ascending CSR position inside every column
        seg = perm[colptr[c]:colptr[c + 1]]
        assert np.all(np.diff(seg) > 0)
    assert np.array_equal(g.colind.numpy()[perm], cols)


def test_sampler_oracle_vs_reference_goldens(golden, oracle):
    z = golden("sampler")
    n = int(z["n"])
    rp, ci, ov = oracle.coo2csr(z["row"], z["col"], z["val"], n)
    assert np.array_equal(rp, z["row_ptr"]) and np.array_equal(ci, z["col_ind"]) and np.array_equal(ov, z["out_val"])
    rp2, perm = oracle.coo2csr_index(z["row"], z["col"], n)
    assert np.array_equal(rp2, z["row_ptr_index"]) and np.array_equal(perm, z["perm"])
    a, b, c, d = oracle.sample_adj(rp, ci, z["seeds"], -1, False)
    assert np.array_equal(a, z["s_indptr"]) and np.array_equal(b, z["s_indices"])
    assert np.array_equal(c, z["s_nodes"]) and np.array_equal(d, z["s_edges"])
    a, b, c, d = oracle.subgraph(rp, ci, z["sub"])
    assert np.array_equal(a, z["g_indptr"]) and np.array_equal(b, z["g_indices"]) and np.array_equal(d, z["g_edges"])


def test_sampler_oracle_vs_ref_build(oracle):
    import os

    if not os.path.exists(os.path.join(os.path.dirname(oracle.__file__), "_ref", "sampler.so")):
        pytest.skip("oracle/_ref not built")
    ref = oracle.ref_sampler()
    g = synth.scaled(3000, 8, seed=9, norm=None)
    indptr, indices = g.rowptr.long(), g.colind.long()
    seeds = torch.randperm(3000, generator=torch.Generator().manual_seed(1))[:256]
    want = ref.sample_adj(indptr, indices, seeds, -1, False)
    got = oracle.sample_adj(indptr, indices, seeds, -1, False)
    for w, g_ in zip(want, got):
        assert np.array_equal(w.numpy(), g_)


def test_edge_softmax_oracle_vs_reference_fallback(golden, oracle):
    z = golden("edge_softmax")
    sm = oracle.edge_softmax_fwd(z["row_indptr"].astype(np.int32), z["values"])
    np.testing.assert_allclose(sm, z["softmax"], rtol=2e-5, atol=1e-7)


def test_gat_oracle_vs_reference_layer(golden, oracle):
    """oracle_gat_fwd (the defined semantics of the fused op) == GATLayer's unfused CPU forward."""
    z = golden("gat_layer")
    h = (z["x"] @ z["W"]).reshape(-1, 4, 8).astype(np.float32)
    h_l = (z["a_l"] * h).sum(-1).astype(np.float32)
    h_r = (z["a_r"] * h).sum(-1).astype(np.float32)
    out = oracle.gat_fwd(z["row_indptr"].astype(np.int32), z["col_indices"].astype(np.int32), h_l, h_r, h, 0.2)
    np.testing.assert_allclose(out.reshape(out.shape[0], -1), z["out"], rtol=2e-4, atol=2e-6)


def test_scatter_max_quirk_region(oracle):
    g = synth.random_csr(30, 30, 4, seed=2, weighted=False)
    x = torch.randn(30, 5, generator=torch.Generator().manual_seed(0)).numpy()
    ref_like, _ = oracle.scatter_max_fwd(g.rowptr, g.colind, x, quirk=True)
    true_max, idx = oracle.scatter_max_fwd(g.rowptr, g.colind, x, quirk=False)
    pos = true_max > np.finfo(np.float32).tiny  # where the reference's FLT_MIN start is harmless
    assert np.array_equal(ref_like[pos], true_max[pos])
    deg = np.diff(g.rowptr.numpy())
    assert np.all(true_max[deg == 0] == 0) and np.all(idx[deg == 0] == -1)


def test_message_ops_oracle_bit_exact_vs_reference(golden, oracle):
    # cogdl/operators/ops.py:4-103 run by the reference itself on an unsorted COO graph (tests/golden/make_golden.py)
    z = golden("message_ops")
    n = int(z["n"])
    for op1 in ("add", "sub", "mul"):
        for op2 in ("sum", "mean"):
            for wkey, w in (("", None), ("_w", z["w"])):
                out = oracle.src_op_e_aggr(op1, op2, z["x"], z["ef"], z["row"], z["col"], n, w=w)
                assert out.tobytes() == z["%s_%s%s" % (op1, op2, wkey)].tobytes(), (op1, op2, wkey)
    out = oracle.src_op_e_aggr("mul", "sum", z["x"], z["es"], z["row"], z["col"], n)
    assert out.tobytes() == z["mul_sum_scalar"].tobytes()
    out = oracle.src_op_e_aggr("add", "sum", None, z["ef"], z["row"], z["col"], n)
    assert out.tobytes() == z["scatter_add"].tobytes()
    assert not z["scatter_add"][n - 20:].any()  # destinations nobody points at stay zero


def test_gat_bwd_oracle_equals_float64_autograd_of_the_unfused_composition(oracle):
    """oracle_gat_bwd (used as the full-size checker of the fused GAT backward) against torch autograd in float64 through
    the layer maths of cogdl/layers/gat_layer.py:73-77; also thread-count independent (rows are independent)."""
    import torch

    from cogdl_amd import synth

    for (m, n_src, h, f, seed) in ((150, 120, 4, 8, 0), (3000, 3000, 8, 8, 1), (90, 200, 1, 41, 2)):
        g = synth.random_csr(m, n_src, 7, seed=seed, weighted=False)
        gen = torch.Generator().manual_seed(seed)
        h_l, h_r = torch.randn(m, h, generator=gen), torch.randn(n_src, h, generator=gen)
        feat, gout = torch.randn(n_src, h, f, generator=gen), torch.randn(m, h, f, generator=gen)
        dd = torch.float64
        l64, r64, f64 = (t.to(dd).requires_grad_() for t in (h_l, h_r, feat))
        row = torch.repeat_interleave(torch.arange(m), g.degrees())
        col = g.colind.long()
        s = torch.nn.functional.leaky_relu(l64[row] + r64[col], 0.2)
        mx = torch.full((m, h), -1e30, dtype=dd).scatter_reduce(0, row.view(-1, 1).expand_as(s), s, "amax")
        e = torch.exp(s - mx[row])
        att = e / torch.zeros(m, h, dtype=dd).index_add_(0, row, e)[row]
        out = torch.zeros(m, h, f, dtype=dd).index_add_(0, row, att.unsqueeze(-1) * f64[col])
        out.backward(gout.to(dd))
        gf, gl, gr = oracle.gat_bwd(g.rowptr, g.colind, h_l, h_r, feat, 0.2, gout, n_src=n_src)
        np.testing.assert_allclose(gf, f64.grad.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gl, l64.grad.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(gr, r64.grad.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(oracle.gat_fwd(g.rowptr, g.colind, h_l, h_r, feat, 0.2), out.detach().numpy(),
                                   rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seeds,fanout,width", [(1024, 10, 128), (128, 10, 100), (8192, 10, 47)])
def test_scatter_max_oracle_against_an_independent_float64_torch_evaluation(oracle, seeds, fanout, width):
    """`oracle_scatter_max_*` restates scatter_max.cu:5-75, which is CUDA-only: NO reference-produced vector exists for it
    and none can be produced here (the row stays "parity unpinned", oracle/README.md).  As far as it CAN be pinned: at
    the block sizes of configs[3] (seeds x fan-out edges into a frontier of sources; MaxAggregator, sage_layer.py:21-29)
    the oracle's maximum equals torch's float64 `scatter_reduce(amax)` over the same edges, its argmax is the FIRST edge
    (CSR order) attaining that maximum -- scatter_max.cu:18-24 updates on `>` only --, and its backward equals
    `index_put_(accumulate=True)` of the routed gradients (scatter_max.cu:44-60) in float64."""
    gen = torch.Generator().manual_seed(seeds + width)
    n_src = seeds * (1 + fanout // 2)
    deg = torch.randint(0, fanout + 1, (seeds,), generator=gen)
    rowptr = torch.zeros(seeds + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(deg, 0).int()
    nnz = int(rowptr[-1])
    colind = torch.randint(0, n_src, (nnz,), generator=gen).int()
    x = torch.randn(n_src, width, generator=gen)
    x[::7] = x[1::7][: x[::7].shape[0]]  # ties between different sources
    out, arg = oracle.scatter_max_fwd(rowptr, colind, x, quirk=False)
    row = torch.repeat_interleave(torch.arange(seeds), deg.long())
    gathered = x.double()[colind.long()]
    want = torch.full((seeds, width), -float("inf"), dtype=torch.float64)
    want = want.scatter_reduce(0, row.view(-1, 1).expand(-1, width), gathered, "amax", include_self=True)
    has = deg > 0
    assert np.array_equal(out[has.numpy()], want[has].float().numpy())
    # first edge in CSR order that attains the maximum
    hit = gathered == want[row]
    pos = torch.where(hit, torch.arange(nnz).view(-1, 1).expand(-1, width), torch.full((1, 1), nnz))
    first = torch.full((seeds, width), nnz, dtype=torch.long).scatter_reduce(0, row.view(-1, 1).expand(-1, width), pos, "amin")
    want_arg = torch.where(first < nnz, colind.long()[first.clamp(max=nnz - 1)], torch.full((1, 1), -1))
    assert np.array_equal(arg[has.numpy()], want_arg[has].int().numpy())
    g = torch.randn(seeds, width, generator=gen)
    got_g = oracle.scatter_max_bwd(g, arg, n_src)
    ok = torch.from_numpy(arg) >= 0
    r, c = torch.nonzero(ok, as_tuple=True)
    want_g = torch.zeros(n_src, width, dtype=torch.float64).index_put_((torch.from_numpy(arg)[ok].long(), c), g.double()[ok], accumulate=True)
    np.testing.assert_allclose(got_g, want_g.numpy(), rtol=1e-6, atol=1e-6)


# ---- pins produced by the reference's OWN CUDA kernels on the MI355X (tests/golden/make_golden_gpu.py) ---------------------
FLT_MIN = np.float32(1.17549435e-38)


def _gpu_cases(z, fields):
    names = sorted({k[: -len("_rowptr")] for k in z if k.endswith("_rowptr")})
    assert names
    return [(n, {f: z["%s_%s" % (n, f)] for f in fields}) for n in names]


def test_scatter_max_oracle_equals_the_reference_cuda_kernel(golden, oracle):
    """scatter_max.npz = outputs of cogdl/operators/scatter_max/scatter_max.cu:5-28 (hipified by the reference's own JIT
    recipe, run on an MI355X; the kernel is one thread per output element, no warp-level operation).
    quirk mode restates the kernel exactly -- out bit-identical everywhere (FLT_MIN where nothing beats it, 0 for empty
    rows), argmax identical wherever the kernel initialised it; the intended operator (quirk off, what the HIP path
    implements) agrees with the reference wherever the row maximum is positive."""
    for name, c in _gpu_cases(golden("scatter_max"), ("rowptr", "colind", "feat", "out", "argmax", "argmax_valid")):
        out_q, arg_q = oracle.scatter_max_fwd(c["rowptr"], c["colind"], c["feat"], quirk=True)
        assert out_q.tobytes() == c["out"].tobytes(), name
        valid = c["argmax_valid"]
        assert valid.any() and np.array_equal(arg_q[valid], c["argmax"][valid]), name
        assert (arg_q[~valid] == -1).all()  # never assigned by the kernel (uninitialised there: not comparable)
        out_t, arg_t = oracle.scatter_max_fwd(c["rowptr"], c["colind"], c["feat"], quirk=False)
        assert np.array_equal(out_t[valid], c["out"][valid]) and np.array_equal(arg_t[valid], c["argmax"][valid]), name
    z = golden("scatter_max")
    assert not z["mixed_k64_argmax_valid"].all() and z["pos_k16_argmax_valid"].sum() > 1000  # both regions are exercised


def test_mhspmm_oracle_equals_the_reference_cuda_kernel(golden, oracle):
    """mhspmm.npz = outputs of cogdl/operators/spmm/multiheadSpmm.cu:6-51 on an MI355X (both launch shapes: mhspmm_1 for
    f < 32, mhspmmSimple otherwise).  Same sequential CSR-order sum per output element; the device compiler contracts
    att * x + acc into an FMA where the oracle (like the CPU reference) rounds the product first: within 1e-5, not bitwise."""
    for name, c in _gpu_cases(golden("mhspmm"), ("rowptr", "colind", "att", "feat", "out")):
        got = oracle.mhspmm(c["rowptr"], c["colind"], c["att"], c["feat"])
        np.testing.assert_allclose(got, c["out"], rtol=1e-5, atol=1e-5, err_msg=name)


def test_mhsddmm_and_mhtranspose_oracles_follow_from_the_pinned_mhspmm(golden, oracle):
    """The reference's mhsddmm / mhtranspose kernels are warp-32 shuffle code / a cuSPARSE call and cannot run on wave64
    (oracle/README.md), so no reference output pins them directly.  What can be had: on the SAME inputs whose mhspmm
    output the reference's kernel produced (mhspmm.npz), oracle_mhsddmm equals the float64 autograd gradient of that
    function with respect to the attention (MHSPMMFunction.backward computes exactly this, cogdl/operators/mhspmm.py:56-66),
    and oracle_mhtranspose is the permutation att[perm] with the oracle's (separately pinned) stable csr2csc perm."""
    for name, c in _gpu_cases(golden("mhspmm"), ("rowptr", "colind", "att", "feat", "out")):
        rowptr, colind = c["rowptr"].astype(np.int64), c["colind"].astype(np.int64)
        row = torch.repeat_interleave(torch.arange(len(rowptr) - 1), torch.from_numpy(np.diff(rowptr)))
        att = torch.from_numpy(c["att"]).double().requires_grad_()
        feat = torch.from_numpy(c["feat"]).double()
        msg = att.unsqueeze(-1) * feat[torch.from_numpy(colind)]
        out = torch.zeros(feat.shape, dtype=torch.float64).index_add_(0, row, msg)
        np.testing.assert_allclose(out.detach().numpy(), c["out"], rtol=1e-5, atol=1e-5, err_msg=name)  # the pinned function
        gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).double()
        (out * gout).sum().backward()
        got = oracle.mhsddmm(c["rowptr"], c["colind"], gout.float().numpy(), c["feat"])
        np.testing.assert_allclose(got, att.grad.numpy(), rtol=1e-5, atol=1e-5, err_msg=name)
        _, _, _, perm = oracle.csr2csc(c["rowptr"], c["colind"], None, n_cols=feat.shape[0])
        assert oracle.mhtranspose(perm, c["att"]).tobytes() == c["att"][perm].tobytes(), name
