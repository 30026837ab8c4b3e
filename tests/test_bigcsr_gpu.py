"""The 64-bit CSR path (cogdl_amd/bigcsr.py, csrc/bigcsr.hip) at sizes the oracle finishes in seconds: the segment
size is forced down (max_edges / tuning key 15) so that small graphs are cut into many segments -- the code that a
3.2e9-edge graph runs, cut where it is cheap to check.  Index outputs bit-exact; csr_spmm bit-exact for rows up to the
long-row threshold (128 edges for these sizes), 1e-5 beyond (re-association at piece borders, as the 32-bit operator).
The full-size graph is tests/test_config5_full_gpu.py."""
import ctypes

import numpy as np
import pytest
import torch

from cogdl_amd import _lib, synth
from cogdl_amd.bigcsr import BigCsr, clear_plans, gather_rows_i64
from cogdl_amd.operators.spmm import csr_spmm_raw, csrspmm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def big(g, max_edges, n_cols=None):
    return BigCsr(g.rowptr.long().to(DEV), g.colind.to(DEV), n_cols=g.n_cols if n_cols is None else n_cols, max_edges=max_edges)


def check_rows(got, want, deg, exact_upto, scale):
    """Rows up to the long-row threshold: bit-exact.  Longer rows (chunk-parallel, re-associated sums): 1e-5 of the
    magnitude of the terms (scale = sum of |w x|), as tests/test_spmm_gpu.py."""
    short = deg <= exact_upto
    assert got[short].tobytes() == want[short].tobytes()
    if (~short).any():
        assert np.all(np.abs(got[~short] - want[~short]) <= 1e-5 * scale[~short] + 1e-6)


@pytest.mark.parametrize("max_edges", [1 << 29, 5000, 700, 64])
@pytest.mark.parametrize("f", [128, 40, 7])
def test_segmented_spmm_matches_oracle(oracle, max_edges, f):
    g = synth.hub_csr(3000, 2500, base_deg=6, seed=3)
    x = rand(g.n_cols, f, seed=1)
    plan = big(g, max_edges)
    rows, edges = plan.segment_rows(), plan.segment_edges()
    assert rows[0] == 0 and rows[-1] == g.num_nodes and edges[-1] == g.nnz
    assert all(b > a for a, b in zip(rows, rows[1:]))
    rp = g.rowptr.long().numpy()
    assert [int(rp[r]) for r in rows] == edges
    seg_edges = np.diff(edges)
    if max_edges < (1 << 29):  # (a row longer than the target owns several cuts, which collapse; at most 64 segments)
        eff = max(max_edges, -(-g.nnz // 64))
        assert plan.n_segments > 1 and seg_edges.max() <= eff + int(g.degrees().max())
    else:
        assert plan.n_segments == 1
    # the rebased row pointers, segment by segment
    r32 = plan.rowptr32.cpu().numpy()
    for s in range(plan.n_segments):
        seg = r32[rows[s] + s: rows[s + 1] + s + 1]
        assert np.array_equal(seg, rp[rows[s]: rows[s + 1] + 1] - edges[s])
    out = plan.spmm(g.weight.to(DEV), x.to(DEV)).cpu().numpy()
    want = oracle.csr_spmm(g.rowptr, g.colind, g.weight, x)
    check_rows(out, want, g.degrees().numpy(), 128, oracle.csr_spmm_abs(g.rowptr, g.colind, g.weight, x))
    # unweighted, and every row sequential (no workspace): bit-exact everywhere
    out = plan.spmm(None, x.to(DEV), split_long_rows=False).cpu().numpy()
    assert out.tobytes() == oracle.csr_spmm(g.rowptr, g.colind, None, x).tobytes()


def test_one_segment_equals_the_32_bit_operator():
    g = synth.arxiv_like(seed=1, topology="rmat")
    x = rand(g.num_nodes, 64, seed=2).to(DEV)
    gd = g.to(DEV)
    plan = BigCsr(gd.rowptr.long(), gd.colind)
    assert plan.n_segments == 1
    assert torch.equal(plan.spmm(gd.weight, x), csr_spmm_raw(gd.rowptr, gd.colind, gd.weight, x))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_segmented_spmm_16_bit(dtype):
    g = synth.random_csr(4000, 3000, 9, seed=5)
    x = rand(g.n_cols, 64, seed=3).to(DEV).to(dtype)
    gd = g.to(DEV)
    w = gd.weight.to(dtype)
    got = big(g, 3000).spmm(w, x)
    assert torch.equal(got, csr_spmm_raw(gd.rowptr, gd.colind, w, x, split_long_rows=False))


@pytest.mark.parametrize("max_edges", [1 << 29, 4096, 300])
@pytest.mark.parametrize("shape", [(3000, 2500), (500, 70000), (9000, 40)])
def test_transpose_i64_bit_exact(oracle, max_edges, shape):
    m, n_cols = shape
    g = synth.hub_csr(m, n_cols, base_deg=5, seed=m)
    plan = big(g, max_edges)
    t, perm, val_t = plan.transpose(g.weight.to(DEV), keep_perm=True)
    colptr, rowind, w_t, perm_ref = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=n_cols)
    assert t.rowptr.dtype == torch.int64 and perm.dtype == torch.int64
    assert np.array_equal(t.rowptr.cpu().numpy(), colptr)
    assert np.array_equal(t.colind.cpu().numpy(), rowind)
    assert np.array_equal(perm.cpu().numpy(), perm_ref)
    assert val_t.cpu().numpy().tobytes() == np.asarray(w_t, dtype=np.float32).tobytes()
    assert torch.equal(gather_rows_i64(perm, g.weight.to(DEV)), val_t)
    assert t.m == n_cols and t.n_cols == m
    # without perm / without values
    t2, perm2, v2 = plan.transpose(None, keep_perm=False)
    assert perm2 is None and v2 is None and torch.equal(t2.colind, t.colind) and torch.equal(t2.rowptr, t.rowptr)
    half = g.weight.to(DEV).half()
    _, _, vh = plan.transpose(half, keep_perm=False)
    assert torch.equal(vh, half[perm])


def test_autograd_through_csrspmm_int64(oracle):
    """csrspmm with an int64 rowptr = the 64-bit path: forward, grad_x (cached 64-bit transpose), grad_w (segmented
    sddmm) against the oracle's csr_spmm / csr2csc / csr_sddmm."""
    clear_plans()
    _lib.hip().cogdl_hip_set_tuning(15, 2000)
    try:
        g = synth.random_csr(3000, 3000, 8, seed=11)
        x, gout = rand(g.n_cols, 32, seed=4), rand(g.num_nodes, 32, seed=5)
        rowptr64, colind = g.rowptr.long().to(DEV), g.colind.to(DEV)
        w_const = g.weight.to(DEV)
        xd = x.to(DEV).requires_grad_()
        out = csrspmm(rowptr64, colind, xd, w_const, True)
        out.backward(gout.to(DEV))
        assert out.detach().cpu().numpy().tobytes() == oracle.csr_spmm(g.rowptr, g.colind, g.weight, x).tobytes()
        colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight)
        assert xd.grad.cpu().numpy().tobytes() == oracle.csr_spmm(colptr, rowind, w_t, gout).tobytes()
        from cogdl_amd.bigcsr import plan_of

        plan = plan_of(rowptr64, colind, g.n_cols)
        assert plan.n_segments > 5 and plan._transposed[1] is None  # constant weights: fused into the transpose, no perm kept
        # a fresh (equal) constant weight tensor per call, as the dispatcher's csr_data.half() makes: the permutation is built
        # ONCE and reused -- no second transpose per new tensor
        for _ in range(2):
            xd.grad = None
            csrspmm(rowptr64, colind, xd, g.weight.to(DEV), True).backward(gout.to(DEV))
            assert xd.grad.cpu().numpy().tobytes() == oracle.csr_spmm(colptr, rowind, w_t, gout).tobytes()
        t_first = plan._transposed[0]
        assert plan._transposed[1] is not None
        csrspmm(rowptr64, colind, xd, g.weight.to(DEV), True).backward(gout.to(DEV))
        assert plan._transposed[0] is t_first
        # learned weights: perm is built, grad_w from the segmented sddmm
        wd = g.weight.to(DEV).requires_grad_()
        xd.grad = None
        out = csrspmm(rowptr64, colind, xd, wd, True)
        out.backward(gout.to(DEV))
        assert xd.grad.cpu().numpy().tobytes() == oracle.csr_spmm(colptr, rowind, w_t, gout).tobytes()
        np.testing.assert_allclose(wd.grad.cpu().numpy(), oracle.csr_sddmm(g.rowptr, g.colind, gout, x), rtol=1e-5, atol=1e-6)
        assert plan._transposed[1] is not None
    finally:
        _lib.hip().cogdl_hip_set_tuning(15, 0)
        clear_plans()


def test_sym_flag_saves_the_transpose_only_for_matrices_that_pass_the_symmetry_test(oracle):
    """`sym` (Graph.is_symmetric(), which the reference trusts blindly: cogdl/operators/spmm.py:63-66) -- the 64-bit path
    verifies it once per (structure, weights): sym-normalised weights on a symmetrised graph reuse the forward plan (no
    transpose is built), row-normalised weights on the SAME structure and a directed graph fall back to the true transpose.
    grad_x is that of the oracle's transpose in all three cases."""
    _lib.hip().cogdl_hip_set_tuning(15, 3000)
    try:
        for name, g, expect in (
            ("sym", synth.scaled(4000, 9, seed=2, topology="rmat", norm="sym"), True),
            ("none", synth.scaled(4000, 9, seed=2, topology="rmat", norm=None), True),
            ("row", synth.scaled(4000, 9, seed=2, topology="rmat", norm="row"), False),
            ("directed", synth.finalize(*synth.uniform_pairs(4000, 20000, 5), 4000, symmetrise=False, norm="sym"), False),
        ):
            clear_plans()
            from cogdl_amd.bigcsr import plan_of

            x, gout = rand(g.n_cols, 24, seed=4), rand(g.num_nodes, 24, seed=5)
            rowptr64, colind = g.rowptr.long().to(DEV), g.colind.to(DEV)
            w = None if g.weight is None else g.weight.to(DEV)
            xd = x.to(DEV).requires_grad_()
            for _ in range(2):
                xd.grad = None
                csrspmm(rowptr64, colind, xd, w, True).backward(gout.to(DEV))
            plan = plan_of(rowptr64, colind, g.n_cols)
            assert plan.n_segments > 3
            assert plan.symmetric_verified(w) is expect, name
            assert (plan._transposed is None) is expect, name
            assert len(plan._sym_checked) == 1
            colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight)
            want_t = oracle.csr_spmm(colptr, rowind, w_t, gout)
            got = xd.grad.cpu().numpy()
            if expect:
                # the reference's own `sym` branch: A again, in A's edge order (cogdl/operators/spmm.py:63-66) -- equal to the
                # transpose's product up to the summation order inside a row (self loops sit at the END of a CSR row here)
                want = oracle.csr_spmm(g.rowptr, g.colind, g.weight, gout)
                check_rows(got, want, g.degrees().numpy(), 128, oracle.csr_spmm_abs(g.rowptr, g.colind, g.weight, gout))
                np.testing.assert_allclose(got, want_t, rtol=1e-5, atol=1e-5)
            else:
                check_rows(got, want_t, np.diff(colptr), 128, oracle.csr_spmm_abs(colptr, rowind, w_t, gout))
            # without the flag: always the transpose
            clear_plans()
            xd.grad = None
            csrspmm(rowptr64, colind, xd, w, False).backward(gout.to(DEV))
            assert plan_of(rowptr64, colind, g.n_cols)._transposed is not None
    finally:
        _lib.hip().cogdl_hip_set_tuning(15, 0)
        clear_plans()


def test_row_schedule_is_a_permutation_per_segment_and_changes_no_bit(monkeypatch):
    """cogdl_hip_csr_spmm_i64_ordered / cogdl_hip_csr_spmm_ordered: the plan's row schedule (decreasing degree inside windows)
    is per segment a permutation of the segment's local row ids; results equal the unordered launch bit for bit -- also for an
    arbitrary permutation through the 32-bit entry."""
    import cogdl_amd.bigcsr as bc

    g = synth.hub_csr(3000, 2500, base_deg=6, seed=3)
    x = rand(g.n_cols, 40, seed=1).to(DEV)
    w = g.weight.to(DEV)
    monkeypatch.setattr(bc, "ROW_WINDOW", 256)
    plan = big(g, 700)
    assert plan.n_segments > 5 and plan.row_order is not None
    rows = plan.segment_rows()
    order = plan.row_order.cpu().numpy()
    deg = g.degrees().numpy()
    for s in range(plan.n_segments):
        loc = order[rows[s]:rows[s + 1]]
        assert np.array_equal(np.sort(loc), np.arange(rows[s + 1] - rows[s]))
        d = deg[rows[s]:rows[s + 1]][loc]
        for w0 in range(0, len(d), 256):  # decreasing inside every window, windows in place
            assert np.all(np.diff(d[w0:w0 + 256]) <= 0) and np.all(loc[w0:w0 + 256] // 256 == w0 // 256)
    out = plan.spmm(w, x)
    monkeypatch.setattr(bc, "ORDER_ROWS", False)
    plain = big(g, 700)
    assert plain.row_order is None and torch.equal(plain.spmm(w, x), out)
    # 32-bit entry, an arbitrary permutation
    lib = _lib.hip()
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    perm = torch.randperm(g.num_nodes, generator=torch.Generator().manual_seed(4)).int().to(DEV)
    ws_bytes = lib.cogdl_hip_csr_spmm_workspace_bytes(g.nnz, 40, 0)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=DEV)
    got = torch.empty(g.num_nodes, 40, device=DEV)
    rc = lib.cogdl_hip_csr_spmm_ordered(rowptr.data_ptr(), colind.data_ptr(), w.data_ptr(), x.data_ptr(), got.data_ptr(), g.num_nodes,
                                        40, g.nnz, 0, 0, perm.data_ptr(), ws.data_ptr(), ws_bytes, None)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(got, csr_spmm_raw(rowptr, colind, w, x))


def test_malformed_row_pointers_are_rejected_at_plan_time():
    """rowptr[0] != 0 or a decreasing rowptr would rebase to negative / wrapped 32-bit pointers."""
    col = torch.zeros(6, dtype=torch.int32, device=DEV)
    for bad in ([1, 2, 4, 6], [0, 4, 2, 6], [0, 7, 6, 6]):
        with pytest.raises(_lib.BackendError):
            BigCsr(torch.tensor(bad, dtype=torch.int64, device=DEV), col, n_cols=3)
    assert BigCsr(torch.tensor([0, 2, 2, 6], dtype=torch.int64, device=DEV), col, n_cols=3).n_segments == 1


def test_empty_and_degenerate_structures():
    z = torch.zeros(1, dtype=torch.int64, device=DEV)
    e = torch.zeros(0, dtype=torch.int32, device=DEV)
    plan = BigCsr(z, e, n_cols=5)
    assert plan.n_segments == 0 and plan.spmm(None, torch.ones(5, 4, device=DEV)).shape == (0, 4)
    rowptr = torch.zeros(11, dtype=torch.int64, device=DEV)  # rows without edges
    plan = BigCsr(rowptr, e, n_cols=5)
    out = plan.spmm(None, torch.ones(5, 4, device=DEV))
    assert out.shape == (10, 4) and float(out.abs().sum()) == 0.0
    t, perm, _ = plan.transpose()
    assert t.m == 5 and int(t.rowptr.abs().sum()) == 0 and perm.numel() == 0
    with pytest.raises(_lib.BackendError):  # rowptr[m] != nnz
        BigCsr(torch.tensor([0, 3], dtype=torch.int64, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))
    with pytest.raises(_lib.BackendError):
        BigCsr(torch.tensor([0, 2], dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))


def test_int32_entry_points_refuse_what_they_cannot_address():
    """The 32-bit entry points return ERANGE above COGDL_HIP_SEGMENT_MAX_EDGES (no launch, pointers never read)."""
    lib = _lib.hip()
    one = torch.zeros(4, dtype=torch.int32, device=DEV)
    f = torch.zeros(4, device=DEV)
    rc = lib.cogdl_hip_csr_spmm(one.data_ptr(), one.data_ptr(), None, f.data_ptr(), f.data_ptr(), 1, 1, 1 << 31, 0, None, 0, None)
    assert rc == 6
    assert ctypes.sizeof(_lib.Segments) == 8 + 2 * 8 * (_lib.MAX_SEGMENTS + 1)
