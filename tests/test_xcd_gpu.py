"""XCD-partitioned column plans (include/cogdl_hip.h: cogdl_hip_vrows; cogdl_amd/xcdplan.py; csrc/rowreduce.h virtual rows).

No reference counterpart (its GE-SpMM kernels know one L2, cogdl/operators/spmm/spmm_kernel.cu:192-512): the plan is a
different EXECUTION ORDER of csr_spmm and of the fused GAT operator, so the checks are the operators' own -- the CPU oracle
(oracle/cogdl_oracle.c: csr_spmm follows spmm_cpu.cpp:24-35, gat_fwd / gat_bwd the GATLayer composition,
cogdl/layers/gat_layer.py:59-86) on small graphs with every kind of row (empty, short = whole, long = eight sub-rows, hubs =
several pieces per owner, rows whose edges all have one owner), with the tolerances of tests/test_config3_gpu.py, plus the
plan's own invariants.  Full-size coverage: tests/test_config3_gpu.py runs the same operators on the Reddit-shaped graph,
where `xcdplan.wanted()` picks the plan by itself.
"""
import numpy as np
import pytest
import torch

from cogdl_amd import synth, xcdplan
from cogdl_amd.operators.fused_gat import fused_gat_dropout_func, edge_dropout_mask
from cogdl_amd.operators.spmm import SPMMFunction, csr_spmm_raw, csr_spmm_xcd_raw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float32: 2e-5, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}
ATOL = {torch.float32: 1e-30, torch.bfloat16: 1e-30, torch.float16: 2.0 ** -24}  # fp16 results below 6e-5 are subnormal


@pytest.fixture(autouse=True)
def forced(monkeypatch):
    monkeypatch.setattr(xcdplan, "MODE", "force")
    xcdplan.XPLANS.clear()
    yield
    xcdplan.XPLANS.clear()


def _graph(kind):
    if kind == "hubs":  # hubs longer than several pieces, medium rows around SPLIT, empty rows
        g = synth.hub_csr(300, 257, base_deg=6, hubs=((3, 129), (4, 1000), (17, 5000), (18, 257), (40, 65), (41, 64), (100, 20000), (299, 700)),
                           seed=5)  # (row 100: ~80 parts -- merged by a whole workgroup)
    elif kind == "one_owner":  # every long row's edges name ONE column: a single owner XCD, seven absent parts
        g = synth.hub_csr(70, 50, base_deg=2, hubs=((0, 300), (5, 66), (69, 513)), seed=2)
        g.colind[g.rowptr[0]:g.rowptr[1]] = 7
        g.colind[g.rowptr[69]:g.rowptr[70]] = 49
    elif kind == "short":  # no long row at all: every slot whole, no records
        g = synth.random_csr(1000, 333, 8, seed=3)
    else:  # rectangular, more rows than columns, R-MAT-like skew
        g = synth.hub_csr(2000, 97, base_deg=30, hubs=((1, 4000), (2, 4001), (1999, 2047)), seed=7)
    return g


KINDS = ["hubs", "one_owner", "short", "rect"]


def _close(got, want, scale, tol, what, atol=1e-30):
    got, want, scale = (np.asarray(a, dtype=np.float64) for a in (got, want, scale))
    err = np.abs(got - want)
    bound = tol * scale + atol
    worst = np.argmax(err / bound)
    assert np.all(err <= bound), "%s: err %.3e > bound %.3e at flat index %d (want %.6e)" % (
        what, err.flat[worst], bound.flat[worst], worst, want.flat[worst])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("split,piece", [(256, 256), (64, 256), (4, 16), (0, 5), (1024, 256)])
def test_plan_invariants(kind, split, piece):
    """Every edge once, CSR order inside a part, owners = the hash, slots of XCD x only in units u % 8 == x, records in
    (row, owner, piece) order."""
    g = _graph(kind)
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    p = xcdplan.build(rowptr, colind, split=split, piece=piece)
    m, nnz = g.num_nodes, colind.numel()
    assert p.n_slots % (xcdplan.UNIT * xcdplan.XCDS) == 0 and p.nnz == nnz and p.m == m
    eid = p.eid.long().cpu()
    assert torch.equal(torch.sort(eid).values, torch.arange(nnz))
    assert torch.equal(p.vcol.cpu(), g.colind[eid])
    vrp, desc = p.vrowptr.long().cpu(), p.vdesc.long().cpu()
    lens = vrp[1:] - vrp[:-1]
    assert int(vrp[0]) == 0 and int(vrp[-1]) == nnz and int(lens.max()) <= max(piece, split)
    row_of_edge = torch.repeat_interleave(torch.arange(m), (g.rowptr[1:] - g.rowptr[:-1]).long())
    slot_of_pos = torch.repeat_interleave(torch.arange(p.n_slots), lens)
    assert torch.equal(desc[slot_of_pos, 0], row_of_edge[eid])  # a slot holds edges of its own row only
    assert bool((lens[desc[:, 0] < 0] == 0).all())  # padding slots are empty
    deg = (g.rowptr[1:] - g.rowptr[:-1]).long()
    unit_xcd = (torch.arange(p.n_slots) // xcdplan.UNIT) % xcdplan.XCDS
    long_edge = deg[row_of_edge[eid]] > split
    own = xcdplan.owner_of(g.colind[eid].long())
    assert torch.equal(unit_xcd[slot_of_pos][long_edge], own[long_edge])  # a long row's edge sits on its column's owner
    # inside a slot the edges keep CSR order
    same = slot_of_pos[1:] == slot_of_pos[:-1]
    assert bool((eid[1:] > eid[:-1])[same].all())
    # a row of at most `split` edges is ONE virtual row, however long (bit-exact rows of fp32 csr_spmm)
    # (a longer row whose edges all have one owner and fit one piece is a single part too)
    is_whole = (desc[:, 0] >= 0) & (desc[:, 1] < 0)
    whole_rows = desc[is_whole, 0]
    assert torch.equal(lens[is_whole], deg[whole_rows])
    assert bool(torch.isin(torch.nonzero(deg <= split).flatten(), whole_rows).all())
    # every row is written exactly once: either one whole slot, or one entry of mrow
    whole = desc[(desc[:, 0] >= 0) & (desc[:, 1] < 0), 0]
    mrow, mptr = p.mrow.long().cpu(), p.mptr.long().cpu()
    assert torch.equal(torch.sort(torch.cat([whole, mrow])).values, torch.arange(m))
    rec = desc[desc[:, 1] >= 0]
    assert p.n_parts == rec.shape[0] == (int(mptr[-1]) if mrow.numel() else 0)
    if rec.shape[0]:
        assert torch.equal(torch.sort(rec[:, 1]).values, torch.arange(p.n_parts))
        rec_row = torch.empty(p.n_parts, dtype=torch.long)
        rec_row[rec[:, 1]] = rec[:, 0]
        assert torch.equal(rec_row, torch.repeat_interleave(mrow, mptr[1:] - mptr[:-1]))
    assert torch.equal(p.big.long().cpu(), torch.nonzero((mptr[1:] - mptr[:-1]) > xcdplan.BIG_PARTS).flatten())
    if kind == "hubs" and split in (64, 256):
        assert p.n_big >= 1


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("k", [1, 2, 6, 16, 40, 64, 128, 200, 602])
def test_csr_spmm_xcd_vs_oracle(oracle, kind, dtype, k):
    g = _graph(kind)
    gen = torch.Generator().manual_seed(k)
    x = torch.randn(g.n_cols, k, generator=gen).to(dtype)
    w = g.weight.to(dtype)
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    plan = xcdplan.build(rowptr, colind)
    want = oracle.csr_spmm_f64(g.rowptr.numpy(), g.colind.numpy(), w.float().numpy(), x.float().numpy())
    scale = oracle.csr_spmm_abs(g.rowptr.numpy(), g.colind.numpy(), w.float().numpy(), x.float().numpy())
    for val in (w, None):
        got = csr_spmm_xcd_raw(plan, None if val is None else val.to(DEV), x.to(DEV))
        assert got.dtype == dtype and got.shape == (g.num_nodes, k)
        if val is None:
            ones = np.ones(g.colind.numel(), dtype=np.float32)
            want_u = oracle.csr_spmm_f64(g.rowptr.numpy(), g.colind.numpy(), ones, x.float().numpy())
            scale_u = oracle.csr_spmm_abs(g.rowptr.numpy(), g.colind.numpy(), ones, x.float().numpy())
            _close(got.float().cpu().numpy(), want_u, scale_u, TOL[dtype], "unweighted", ATOL[dtype])
        else:
            _close(got.float().cpu().numpy(), want, scale, TOL[dtype], "weighted", ATOL[dtype])
    # out += A x, and the launch is deterministic
    base = torch.randn(g.num_nodes, k, generator=gen).to(dtype).to(DEV)
    acc = csr_spmm_xcd_raw(plan, w.to(DEV), x.to(DEV), out=base.clone())
    _close(acc.float().cpu().numpy(), want + base.float().cpu().numpy().astype(np.float64),
           scale + np.abs(base.float().cpu().numpy()), 2 * TOL[dtype], "accumulate", ATOL[dtype])
    again = csr_spmm_xcd_raw(plan, w.to(DEV), x.to(DEV))
    assert torch.equal(again, csr_spmm_xcd_raw(plan, w.to(DEV), x.to(DEV)))


def test_csr_spmm_xcd_short_rows_are_bit_exact(oracle):
    """Rows of at most SPLIT edges stay whole: sequential CSR order, bit-identical to the reference loop in fp32."""
    g = _graph("short")
    x = torch.randn(g.n_cols, 64)
    plan = xcdplan.build(g.rowptr.to(DEV), g.colind.to(DEV))
    assert plan.n_parts == 0
    got = csr_spmm_xcd_raw(plan, g.weight.to(DEV), x.to(DEV))
    want = oracle.csr_spmm(g.rowptr.numpy(), g.colind.numpy(), g.weight.numpy(), x.numpy())
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_spmm_function_takes_the_plan_in_both_directions(oracle, dtype, monkeypatch):
    """SPMMFunction under COGDL_AMD_XCD=force: forward over the CSR plan, grad_x over the plan of the transpose (edge weights
    stay in CSR order: the CSC plan's eid is the transpose's perm composed with the plan)."""
    g = _graph("hubs")
    calls = []
    import cogdl_amd.operators.spmm as spmm_mod

    real = spmm_mod.csr_spmm_xcd_raw
    monkeypatch.setattr(spmm_mod, "csr_spmm_xcd_raw", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = torch.randn(g.n_cols, 32).to(dtype).to(DEV).requires_grad_()
    w = g.weight.to(dtype).to(DEV)
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    out = SPMMFunction.apply(rowptr, colind, x, w, False)
    gout = torch.randn(g.num_nodes, 32).to(dtype).to(DEV)
    out.backward(gout)
    assert len(calls) == 2
    rp, ci, wn = g.rowptr.numpy(), g.colind.numpy(), w.float().cpu().numpy()
    xn = x.detach().float().cpu().numpy()
    _close(out.detach().float().cpu().numpy(), oracle.csr_spmm_f64(rp, ci, wn, xn), oracle.csr_spmm_abs(rp, ci, wn, xn),
           TOL[dtype], "forward")
    # grad_x = A^T gout: through an independently built transpose (scipy)
    import scipy.sparse as sp

    at = sp.csr_matrix((wn.astype(np.float64), ci, rp), shape=(g.num_nodes, g.n_cols)).T.tocsr()
    gn = gout.float().cpu().numpy().astype(np.float64)
    _close(x.grad.float().cpu().numpy(), at @ gn, abs(at) @ np.abs(gn), TOL[dtype], "grad_x")


SHAPES = [(8, 8), (1, 41), (4, 16), (2, 64), (1, 8), (3, 5)]


@pytest.mark.parametrize("kind", ["hubs", "one_owner", "rect"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("h,f", SHAPES)
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_fused_gat_xcd_vs_oracle(oracle, kind, dtype, h, f, p):
    g = _graph(kind)
    if g.num_nodes != g.n_cols:  # the operator's graphs are square
        n = max(g.num_nodes, g.n_cols)
        deg = torch.zeros(n, dtype=torch.long)
        deg[:g.num_nodes] = (g.rowptr[1:] - g.rowptr[:-1]).long()
        rp = torch.zeros(n + 1, dtype=torch.long)
        torch.cumsum(deg, 0, out=rp[1:])
        g = synth.CSRGraph(rp.int(), g.colind, g.weight, n, n)
    n = g.num_nodes
    tol, seed = TOL[dtype], 1234 + h
    gen = torch.Generator().manual_seed(100 * h + f)
    ar0, ac0 = torch.randn(n, h, generator=gen), torch.randn(n, h, generator=gen)
    feat0 = torch.randn(n, h, f, generator=gen).to(dtype)
    gout = torch.randn(n, h, f, generator=gen).to(dtype)
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    ar, ac, ft = (t.to(DEV).requires_grad_() for t in (ar0, ac0, feat0))
    out = fused_gat_dropout_func(ar, ac, rowptr, colind, 0.2, ft, p, seed)
    out.backward(gout.to(DEV))
    torch.cuda.synchronize()
    drop = edge_dropout_mask(colind.numel(), h, p, seed, DEV).cpu().numpy() if p > 0 else None
    rpn, cin = g.rowptr.numpy(), g.colind.numpy()
    fh, gh = feat0.float().numpy(), gout.float().numpy()
    want = oracle.gat_fwd(rpn, cin, ar0.numpy(), ac0.numpy(), fh, 0.2, drop=drop)
    scale = oracle.gat_fwd(rpn, cin, ar0.numpy(), ac0.numpy(), np.abs(fh), 0.2, drop=drop)
    _close(out.detach().float().cpu().numpy(), want, scale, tol, "forward")
    gf, gl, gr, sf, sl, sr = oracle.gat_bwd(rpn, cin, ar0.numpy(), ac0.numpy(), fh, 0.2, gh, scales=True, drop=drop)
    k = 4 if dtype != torch.float32 else 1
    _close(ft.grad.float().cpu().numpy(), gf, sf, tol, "grad_feat")
    _close(ar.grad.cpu().numpy(), gl, sl, tol * k, "grad_attn_row")
    _close(ac.grad.cpu().numpy(), gr, sr, tol * k, "grad_attn_col")


def test_fused_gat_xcd_kernels_are_the_ones_that_ran(monkeypatch):
    """The plan entries return EUNSUPPORTED for the shapes they decline (the caller falls back silently): make sure the
    shapes of configs[2] do NOT fall back."""
    from cogdl_amd import _lib

    g = _graph("hubs")
    lib = _lib.hip()
    n = g.num_nodes
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    plan = xcdplan.build(rowptr, colind)
    for h, f in ((8, 8), (1, 48)):
        feat = torch.randn(n, h, f, device=DEV).bfloat16()
        ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
        out = torch.empty_like(feat)
        emax, esum = torch.empty(n, h, device=DEV), torch.empty(n, h, device=DEV)
        ws, wsb = _lib.workspace("cogdl_hip_gat_fwd_xcd_workspace_bytes", torch.device(DEV), plan.n_parts, h, f, 2)
        for p in (0.0, 0.5):
            rc = lib.cogdl_hip_gat_fwd_xcd(plan.ref(), _lib.ptr(ar), _lib.ptr(ac), _lib.ptr(feat), 0.2, p, 7, _lib.ptr(out),
                                           _lib.ptr(emax), _lib.ptr(esum), n, h, f, 2, _lib.ptr(ws), wsb, None)
            assert rc == 0, (h, f, p, rc)
    torch.cuda.synchronize()


def test_fused_gat_under_hipgraph_capture_replays_the_recorded_plans():
    """cogdl_amd.graphs.capture: the recorded eager run decides (plan.taped_choice) and builds the plans, the capture takes them
    from the tape -- nothing is hashed or read back while capturing -- and the captured step keeps them alive.  The replays
    (also on new inputs written into the static tensors) equal the eager operator bit for bit: same plans, same kernels."""
    from cogdl_amd import graphs, plan as plan_mod

    g = synth.hub_csr(400, 400, base_deg=6, hubs=((3, 129), (17, 5000), (100, 20000), (399, 700)), seed=11)
    n, h, f = 400, 8, 8
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    gen = torch.Generator().manual_seed(3)
    ar, ac = (torch.randn(n, h, generator=gen).to(DEV).requires_grad_() for _ in range(2))
    ft = torch.randn(n, h, f, generator=gen).bfloat16().to(DEV).requires_grad_()
    gout = torch.randn(n, h, f, generator=gen).bfloat16().to(DEV)
    res = [torch.zeros(n, h, f, device=DEV), torch.zeros(n, h, f, device=DEV), torch.zeros(n, h, device=DEV), torch.zeros(n, h, device=DEV)]

    def step():
        for t in (ar, ac, ft):
            t.grad = None
        out = fused_gat_dropout_func(ar, ac, rowptr, colind, 0.2, ft, 0.0, 0)
        out.backward(gout)
        for r, t in zip(res, (out.detach(), ft.grad, ar.grad, ac.grad)):
            r.copy_(t)
        return res

    def eager():
        step()
        torch.cuda.synchronize()
        return [r.clone() for r in res]

    want0 = eager()
    tapes, set_tape = [], plan_mod.set_tape

    def spy(tape):
        if tape is not None and tape not in tapes:
            tapes.append(tape)
        set_tape(tape)

    plan_mod.set_tape = spy
    try:
        captured = graphs.capture(step, warmup=2)
    finally:
        plan_mod.set_tape = set_tape
    (tape,) = tapes
    assert [k for k, _ in tape.choices] == ["fused_gat.forward", "fused_gat.backward"]
    assert isinstance(tape.choices[0][1], xcdplan.XcdPlan) and all(isinstance(p, xcdplan.XcdPlan) for p in tape.choices[1][1])
    xcdplan.XPLANS.clear()
    plan_mod.PLANS.clear()
    del tape, tapes
    for r in res:
        r.zero_()
    captured()
    torch.cuda.synchronize()
    for got, want in zip(res, want0):
        assert torch.equal(got, want)
    with torch.no_grad():  # new inputs in the static tensors
        ft.copy_(torch.randn(n, h, f, generator=gen).bfloat16())
        ar.copy_(torch.randn(n, h, generator=gen))
    captured()
    torch.cuda.synchronize()
    got1 = [r.clone() for r in res]
    want1 = eager()
    for got, want in zip(got1, want1):
        assert torch.equal(got, want)


def test_csrspmm_fp32_on_the_reddit_shaped_graph_takes_the_plan_and_keeps_its_exact_rows(oracle, monkeypatch):
    """fp32 csr_spmm over a hub-heavy structure and a cache-sized table (auto mode): the plan is cut at the exact-row bound of
    the ordinary path, so every row of at most cogdl_hip_exact_row_edges(nnz) edges is still BIT-identical to the reference loop
    (spmm_cpu.cpp:24-35, the oracle) and the longer ones are within the fp32 tolerance; grad_x over the plan of the transpose
    equals the ordinary path's within the same tolerance."""
    from cogdl_amd import _lib
    import cogdl_amd.operators.spmm as spmm_mod

    monkeypatch.setattr(xcdplan, "MODE", "auto")
    g = synth.reddit_like(seed=0, device=DEV, norm="sym")
    n, k = g.num_nodes, 32
    calls = []
    real = spmm_mod.csr_spmm_xcd_raw
    monkeypatch.setattr(spmm_mod, "csr_spmm_xcd_raw", lambda *a, **kw: (calls.append(a[0]), real(*a, **kw))[1])
    x = torch.randn(n, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).requires_grad_()
    gout = torch.randn(n, k, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    out = SPMMFunction.apply(g.rowptr, g.colind, x, g.weight, False)
    out.backward(gout)
    bound = int(_lib.hip().cogdl_hip_exact_row_edges(g.nnz))
    assert len(calls) == 2 and bound >= 128
    deg = g.degrees().cpu().numpy()
    rp, ci, w = g.rowptr.cpu().numpy(), g.colind.cpu().numpy(), g.weight.cpu().numpy()
    xh = x.detach().cpu().numpy()
    want = oracle.csr_spmm(rp, ci, w, xh, nthreads=16)
    got = out.detach().cpu().numpy()
    short = deg <= bound
    assert short.sum() > 100_000 and (~short).sum() > 1000
    assert np.array_equal(got[short], want[short])
    scale = oracle.csr_spmm_abs(rp, ci, w, xh)
    _close(got, oracle.csr_spmm_f64(rp, ci, w, xh), scale, TOL[torch.float32], "forward (long rows re-associated)")
    gx_plan = x.grad.clone()
    x.grad = None
    monkeypatch.setattr(xcdplan, "MODE", "off")
    SPMMFunction.apply(g.rowptr, g.colind, x, g.weight, False).backward(gout)
    assert len(calls) == 2
    tol = 2e-5 * float(gout.abs().max()) * float(g.weight.abs().max()) * float(deg.max())
    assert float((gx_plan - x.grad).abs().max()) <= tol


def test_skewed_structures_of_any_size_take_a_plan_once_their_fingerprint_is_known(oracle, monkeypatch):
    """xcdplan.ordered_wanted: the arxiv-sized R-MAT graph is far below `wanted()`'s size, but skewed -- the BACKWARD pass (which
    has the structure's key anyway) walks the transpose's plan; the forward pass does too once the fingerprint is memoised with
    the structure (install(structure_memo=True): here the memo record is attached by hand).  Rows up to the exact-row bound are
    bit-identical to the ordinary launch; the uniform graph of the same size keeps the ordinary launch throughout."""
    import cogdl_amd.operators.spmm as spmm_mod
    from cogdl_amd import _lib
    from cogdl_amd.structure_memo import StructureMemo

    monkeypatch.setattr(xcdplan, "MODE", "auto")
    xcdplan._SKEW.clear()
    calls = []
    real = spmm_mod.csr_spmm_xcd_raw
    monkeypatch.setattr(spmm_mod, "csr_spmm_xcd_raw", lambda *a, **k: (calls.append(a[0]), real(*a, **k))[1])
    for topo, expect in (("rmat", True), ("uniform", False)):
        g = synth.arxiv_like(seed=0, topology=topo)
        rowptr, colind, w = g.rowptr.to(DEV), g.colind.to(DEV), g.weight.to(DEV)
        x = torch.randn(g.num_nodes, 40, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).requires_grad_()
        gout = torch.randn(g.num_nodes, 40, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        plain = csr_spmm_raw(rowptr, colind, w, x.detach())
        del calls[:]
        out = SPMMFunction.apply(rowptr.clone(), colind.clone(), x, w, False)  # fresh tensors: the hash is in flight, the forward stays ordinary
        assert not calls and torch.equal(out, plain)
        out.backward(gout)
        assert not calls  # first sighting of the structure: a one-off structure must not pay a plan build
        gx_plain = x.grad.clone()
        x.grad = None
        out = SPMMFunction.apply(rowptr.clone(), colind.clone(), x, w, False)  # (new objects again: no identity memo)
        assert not calls and torch.equal(out, plain)
        out.backward(gout)
        assert len(calls) == (1 if expect else 0), topo  # second sighting: the transpose's plan
        gx_first = x.grad.clone()
        scale_t = gx_plain.abs().max()
        assert bool(((gx_first - gx_plain).abs() <= 2e-5 * scale_t).all())
        # ... memoised structure: both directions
        memo = StructureMemo()
        memo.rowptr32, memo.colind32 = rowptr, colind
        rowptr._cogdl_amd_struct = memo
        del calls[:]
        x.grad = None
        out = SPMMFunction.apply(rowptr, colind, x, w, False)
        out.backward(gout)
        assert len(calls) == (2 if expect else 0), topo
        if expect:
            assert calls[0].n_parts > 0 and calls[0].m == g.num_nodes
            exact = int(_lib.hip().cogdl_hip_exact_row_edges(g.nnz))
            short = (g.degrees() <= exact).to(DEV)
            assert torch.equal(out[short], plain[short])  # whole virtual rows: the reference's sequential order
            scale = csr_spmm_raw(rowptr, colind, w.abs(), x.detach().abs())
            assert bool(((out - plain).abs() <= 2e-5 * scale + 1e-30).all())
            assert torch.equal(x.grad, gx_first)  # the same plan of the transpose as the un-memoised backward
        del rowptr._cogdl_amd_struct
    xcdplan._SKEW.clear()


def test_identity_memo_of_fingerprints():
    """plan.fingerprint_of: the very same index tensor OBJECTS are hashed once; an in-place edit, a clone or other objects with
    the same contents are hashed again (and agree on the key); known_fingerprint never hashes."""
    from cogdl_amd import plan as _plan

    _plan.clear_identity_memo()
    g = synth.random_csr(500, 400, 6, seed=1)
    rowptr, colind = g.rowptr.to(DEV), g.colind.to(DEV)
    assert _plan.known_fingerprint(rowptr, colind, 400) is None
    fp = _plan.fingerprint_of(rowptr, colind, 400)
    assert _plan.fingerprint_of(rowptr, colind, 400) is fp and _plan.known_fingerprint(rowptr, colind, 400) is fp
    assert _plan.fingerprint_of(rowptr, colind, 401) is not fp  # another column count: another key
    other = _plan.fingerprint_of(rowptr.clone(), colind.clone(), 400)
    assert other is not fp and other.key() == fp.key()
    colind[0] = (int(colind[0]) + 1) % 400  # in place: the version counter moves, the memo misses, the key changes
    fp2 = _plan.fingerprint_of(rowptr, colind, 400)
    assert fp2 is not fp and fp2.key() != fp.key()
    del rowptr, colind
    import gc

    gc.collect()
    assert all(h[0]() is None or h[1]() is not None for h in _plan._IDENT.values())  # weak references: nothing kept alive
    _plan.clear_identity_memo()


def test_wanted_rule():
    xcdplan.MODE = "auto"
    n, nnz = synth.REDDIT_NODES, 114_848_857
    assert xcdplan.wanted(n, nnz, n, 128)  # configs[2], bf16 H x F = 64
    assert not xcdplan.wanted(n, nnz, n, 96)  # its output layer (41 -> 48 columns, one head): measured no gain
    assert not xcdplan.wanted(169_343, 2_501_719, 169_343, 512)  # arxiv: 15 edges per row
    assert not xcdplan.wanted(111_059_956, 1_615_685_872, 111_059_956, 512)  # papers: the table is 57 GB
    xcdplan.MODE = "force"
    assert not xcdplan.wanted(1 << 24, 1 << 30, 1 << 24, 64)  # beyond the plan kernels' 24-bit row ids, even when forced
    xcdplan.MODE = "off"
    assert not xcdplan.wanted(n, nnz, n, 128)
