"""GPU parity of cogdl_amd.operators.ops (fused cogdl_hip_gspmm kernel, through the C ABI) against
  * the reference's own outputs (tests/golden/message_ops.npz): bit-exact -- per output element the kernel adds the
    edges in the caller's COO order with the reference's roundings;
  * the oracle (oracle_src_op_e_aggr, pinned against the same goldens) on larger seeded graphs: bit-exact while every
    destination has at most `long-row threshold` edges, 1e-5 relative (re-association only) for hub destinations.
"""
import types

import numpy as np
import pytest
import torch

from cogdl_amd import _lib
from cogdl_amd.operators import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(row, col, w=None):
    g = types.SimpleNamespace()
    g.edge_index = (row, col)
    g.edge_weight = w
    return g


def _coo(n, e, seed, sort=False, hub=None):
    gen = torch.Generator().manual_seed(seed)
    row = torch.randint(0, max(1, n - n // 8), (e,), generator=gen)  # the last n/8 destinations stay empty
    col = torch.randint(0, n, (e,), generator=gen)
    if hub:
        for node, cnt in hub:
            row[torch.randperm(e, generator=gen)[:cnt]] = node
    if sort:
        row, order = torch.sort(row, stable=True)
        col = col[order]
    return row, col


def test_goldens_bit_exact(golden):
    z = golden("message_ops")
    n = int(z["n"])
    t = {k: torch.from_numpy(z[k]).to(DEV) for k in ("row", "col", "x", "ef", "es", "w", "G")}
    g = _graph(t["row"], t["col"], t["w"])
    row_before = t["row"].clone()
    for op1 in ("add", "sub", "mul"):
        for op2 in ("sum", "mean"):
            fn = getattr(ops, "s_%s_e_%s" % (op1, op2))
            assert fn(g, t["x"], t["ef"]).cpu().numpy().tobytes() == z["%s_%s" % (op1, op2)].tobytes(), (op1, op2)
            assert fn(g, t["x"], t["ef"], weight=True).cpu().numpy().tobytes() == \
                z["%s_%s_w" % (op1, op2)].tobytes(), (op1, op2, "w")
    assert ops.s_mul_e_sum(g, t["x"], t["es"]).cpu().numpy().tobytes() == z["mul_sum_scalar"].tobytes()
    assert ops.scatter_add(t["ef"], t["row"], n).cpu().numpy().tobytes() == z["scatter_add"].tobytes()
    assert ops.op_aggr("sum", t["ef"], t["row"], n).cpu().numpy().tobytes() == z["scatter_add"].tobytes()
    assert torch.equal(t["row"], row_before)  # the caller's edge list is never reordered
    # gradients (reference autograd through the torch composition)
    xg, eg = t["x"].clone().requires_grad_(), t["ef"].clone().requires_grad_()
    # The fused backward (gspmm over the source-sorted view + the per-edge kernel) keeps autograd's roundings and adds the
    # edges of a source in edge order, as the reference's CPU index_add_ does: the reference's gradients BIT FOR BIT
    # (the sum over the columns of a scalar edge feature has no defined order in torch: 1e-5 there).
    (ops.s_mul_e_mean(g, xg, eg, weight=True) * t["G"]).sum().backward()
    assert xg.grad.cpu().numpy().tobytes() == z["grad_x_mul_mean_w"].tobytes()
    assert eg.grad.cpu().numpy().tobytes() == z["grad_e_mul_mean_w"].tobytes()
    xg, sg = t["x"].clone().requires_grad_(), t["es"].clone().requires_grad_()
    (ops.s_sub_e_sum(g, xg, sg) * t["G"]).sum().backward()
    assert xg.grad.cpu().numpy().tobytes() == z["grad_x_sub_sum_scalar"].tobytes()
    np.testing.assert_allclose(sg.grad.cpu().numpy(), z["grad_e_sub_sum_scalar"], rtol=1e-5, atol=1e-5)


def test_fused_backward_has_no_edge_sized_temporaries():
    """s_mul_e_sum forward + backward at the arxiv-shaped size, F = 64: with the edge features constant (the GCN-style use:
    only grad_x wanted) the backward allocates NOTHING of size [E, F] -- the torch composition it replaces held three
    (g_msg, d_src and the gathered source rows); peak memory over the step is bounded by the inputs + two [N, F] results."""
    n, e, k = 169_343, 2_332_486, 64
    gen = torch.Generator(device=DEV).manual_seed(0)
    row = torch.randint(0, n, (e,), device=DEV, generator=gen)
    col = torch.randint(0, n, (e,), device=DEV, generator=gen)
    x = torch.randn(n, k, device=DEV, generator=gen).requires_grad_()
    ef = torch.randn(e, k, device=DEV, generator=gen)
    G = torch.randn(n, k, device=DEV, generator=gen)
    g = _graph(row, col, None)
    ops.s_mul_e_sum(g, x, ef).backward(G)  # (plans and workspaces exist after this)
    x.grad = None
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    ops.s_mul_e_sum(g, x, ef).backward(G)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    edge_tensor = e * k * 4
    assert extra < 0.5 * edge_tensor, "backward allocated %.1f MB beyond the step's inputs ([E, F] = %.1f MB)" % (extra / 1e6, edge_tensor / 1e6)
    # and the result is the torch composition's
    xb = x.detach().clone().requires_grad_()
    (torch.zeros(n, k, device=DEV).index_add_(0, row, xb[col] * ef) * G).sum().backward()
    assert torch.allclose(x.grad, xb.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", [1, 3, 8, 40, 64, 100, 128, 300])
@pytest.mark.parametrize("sort", [False, True])
def test_fused_vs_oracle_bit_exact(oracle, k, sort):
    n, e = 3000, 24000
    row, col = _coo(n, e, seed=k, sort=sort)
    gen = torch.Generator().manual_seed(100 + k)
    x, ef, w = torch.randn(n, k, generator=gen), torch.randn(e, k, generator=gen), torch.rand(e, generator=gen)
    g = _graph(row.to(DEV), col.to(DEV), w.to(DEV))
    for op1 in ("add", "sub", "mul"):
        for op2 in ("sum", "mean"):
            for weight in (False, True):
                got = getattr(ops, "s_%s_e_%s" % (op1, op2))(g, x.to(DEV), ef.to(DEV), weight=weight)
                want = oracle.src_op_e_aggr(op1, op2, x, ef, row, col, n, w=w if weight else None)
                assert got.cpu().numpy().tobytes() == want.tobytes(), (op1, op2, weight)
    got = ops.scatter_add(ef.to(DEV), g.edge_index[0], n)
    assert got.cpu().numpy().tobytes() == oracle.src_op_e_aggr("add", "sum", None, ef, row, col, n).tobytes()
    got = ops.op_aggr("mean", ef.to(DEV), g.edge_index[0].view(-1, 1), n)  # [E, 1] index like scatter_add's expand
    assert got.cpu().numpy().tobytes() == oracle.src_op_e_aggr("add", "mean", None, ef, row, col, n).tobytes()
    if k > 1:
        es = torch.randn(e, generator=gen)
        got = ops.s_sub_e_sum(g, x.to(DEV), es.to(DEV))
        assert got.cpu().numpy().tobytes() == oracle.src_op_e_aggr("sub", "sum", x, es, row, col, n).tobytes()


@pytest.mark.parametrize("k", [16, 128])
def test_hub_destinations(oracle, k):
    """Destinations with 10^3..10^4 edges take the chunk-parallel path: deterministic, equal up to re-association."""
    n, e = 5000, 120000
    row, col = _coo(n, e, seed=7, hub=((3, 129), (4, 20000), (17, 3000), (18, 700)))
    gen = torch.Generator().manual_seed(k)
    x, ef, w = torch.randn(n, k, generator=gen), torch.randn(e, k, generator=gen), torch.rand(e, generator=gen)
    g = _graph(row.to(DEV), col.to(DEV), w.to(DEV))
    thresh = _lib.hip().cogdl_hip_exact_row_edges(e)
    deg = torch.bincount(row, minlength=n)
    short = (deg <= thresh).numpy()
    assert (~short).sum() >= 3
    for op1, op2 in (("mul", "sum"), ("add", "mean")):
        got = getattr(ops, "s_%s_e_%s" % (op1, op2))(g, x.to(DEV), ef.to(DEV), weight=True)
        again = getattr(ops, "s_%s_e_%s" % (op1, op2))(g, x.to(DEV), ef.to(DEV), weight=True)
        assert torch.equal(got, again)  # no atomics: run-to-run identical
        want = oracle.src_op_e_aggr(op1, op2, x, ef, row, col, n, w=w)
        got = got.cpu().numpy()
        assert got[short].tobytes() == want[short].tobytes()
        msg_abs = oracle.src_op_e_aggr(op1, op2, x.abs(), ef.abs(), row, col, n, w=w)  # scale of the summands
        assert np.all(np.abs(got - want) <= 1e-5 * np.maximum(msg_abs, 1e-30))


def test_plan_is_memoised_and_graph_untouched():
    ops.clear_plans()
    row, col = _coo(500, 4000, seed=1)
    g = _graph(row.to(DEV), col.to(DEV), None)
    x, ef = torch.randn(500, 8, device=DEV), torch.randn(4000, 8, device=DEV)
    a = ops.s_add_e_sum(g, x, ef)
    assert len(ops._PLANS) == 1
    b = ops.s_add_e_mean(g, x, ef)
    assert len(ops._PLANS) == 1 and a.shape == b.shape
    g.edge_index[0][0] = 7  # an in-place edit bumps the version: a new plan, not a stale one
    ops.s_add_e_sum(g, x, ef)
    assert len(ops._PLANS) == 2


def test_out_of_range_destination_raises():
    data = torch.ones(4, 2, device=DEV)
    with pytest.raises(_lib.BackendError):
        ops.scatter_add(data, torch.tensor([0, 1, 9, 2], device=DEV), 3)


def test_gradcheck_like_against_torch_composition():
    """Backward of the fused op == autograd through the reference's unfused torch expression (fp64-free: compare with
    the same composition evaluated by torch on the GPU, 1e-5 relative)."""
    n, e, k = 800, 6000, 20
    row, col = _coo(n, e, seed=11)
    row, col = row.to(DEV), col.to(DEV)
    gen = torch.Generator().manual_seed(5)
    x0, e0, w0 = (torch.randn(n, k, generator=gen).to(DEV), torch.randn(e, k, generator=gen).to(DEV),
                  torch.rand(e, generator=gen).to(DEV))
    G = torch.randn(n, k, generator=gen).to(DEV)
    for op1 in ("add", "sub", "mul"):
        xa, ea, wa = (t.clone().requires_grad_() for t in (x0, e0, w0))
        (ops.src_op_e_aggr_coo(op1, "mean", xa, ea, row, col, data=wa) * G).sum().backward()
        xb, eb, wb = (t.clone().requires_grad_() for t in (x0, e0, w0))
        msg = ops.op_src_edge(op1, xb[col], eb) * wb.view(-1, 1)
        out = torch.zeros(n, k, device=DEV).index_add_(0, row, msg)
        deg = torch.bincount(row, minlength=n).float().clamp(min=1).view(-1, 1)
        ((out / deg) * G).sum().backward()
        for a, b in ((xa, xb), (ea, eb), (wa, wb)):
            assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-5), op1


def test_backward_with_strided_row_and_col_of_an_edge_list():
    """Regression: row = ei[:, 0], col = ei[:, 1] of an [E, 2] edge list are both NON-contiguous.
    The backward's per-edge kernel used to take the address of two temporary `.contiguous()` copies; the second copy
    reused the block the first had just given back, so the kernel read the source ids as destinations.  Gradients must
    equal those of the same call on contiguous index tensors, bit for bit."""
    n, e, k = 700, 9000, 24
    row, col = _coo(n, e, seed=21)
    ei = torch.stack([row, col], dim=1).to(DEV)  # [E, 2]: both columns have stride 2
    r_s, c_s = ei[:, 0], ei[:, 1]
    assert not r_s.is_contiguous() and not c_s.is_contiguous()
    r_c, c_c = r_s.contiguous(), c_s.contiguous()
    gen = torch.Generator().manual_seed(8)
    x0, e0, w0 = (torch.randn(n, k, generator=gen).to(DEV), torch.randn(e, k, generator=gen).to(DEV),
                  torch.rand(e, generator=gen).to(DEV))
    G = torch.randn(n, k, generator=gen).to(DEV)
    for op1 in ("add", "mul"):
        grads = []
        for r, c in ((r_c, c_c), (r_s, c_s)):
            ops.clear_plans()
            xa, ea, wa = (t.clone().requires_grad_() for t in (x0, e0, w0))
            torch.cuda.empty_cache()
            (ops.src_op_e_aggr_coo(op1, "sum", xa, ea, r, c, data=wa) * G).sum().backward()
            grads.append((xa.grad, ea.grad, wa.grad))
        for a, b in zip(*grads):
            assert torch.equal(a, b), op1


def test_fresh_source_ids_on_a_fixed_destination_list(oracle):
    """Regression (ADVICE medium): `col` resampled every step against a persistent `row` tensor (negative sampling).
    Each step's `col` dies with its frame and the allocator recycles its address for the next one; the plan's sorted
    source-id memo must not serve the previous step's ids."""
    ops.clear_plans()
    n, e = 400, 3000
    row, _ = _coo(n, e, seed=3)
    row_d = row.to(DEV)
    x = torch.randn(n, 8, generator=torch.Generator().manual_seed(1))
    ef = torch.randn(e, 8, generator=torch.Generator().manual_seed(2))
    xd, efd = x.to(DEV), ef.to(DEV)

    def step(k):
        col = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(50 + k))
        got = ops.s_mul_e_sum(_graph(row_d, col.to(DEV)), xd, efd).cpu().numpy()  # col.to(DEV) is freed on return
        want = oracle.src_op_e_aggr("mul", "sum", x, ef, row, col, n)
        assert got.tobytes() == want.tobytes(), k

    for k in range(4):
        step(k)
    assert len(ops._PLANS) == 1  # one destination plan served all four steps


def test_gpu_calls_outside_the_fused_kernel_take_the_torch_route_loudly():
    """The aggregating operators are torch compositions in the reference (cogdl/operators/ops.py:4-52); a GPU call whose
    arguments the fused kernel does not cover (half precision here) runs those expressions -- with ONE TorchRouteWarning
    per (operator, reason), never silently -- and CPU calls (the reference's own path) stay quiet."""
    import warnings

    from cogdl_amd.operators import ops

    ops._ROUTE_NOTED.clear()
    n, e = 50, 400
    gen = torch.Generator().manual_seed(0)
    row, col = torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)
    x, ef = torch.randn(n, 8, generator=gen), torch.randn(e, 8, generator=gen)
    want = ops.src_op_e_aggr_coo("mul", "sum", x, ef, row, col)  # CPU: quiet
    with warnings.catch_warnings():
        warnings.simplefilter("error", ops.TorchRouteWarning)
        ops.src_op_e_aggr_coo("mul", "sum", x, ef, row, col)
        got32 = ops.src_op_e_aggr_coo("mul", "sum", x.to(DEV), ef.to(DEV), row.to(DEV), col.to(DEV))  # fused: quiet
    assert torch.allclose(got32.cpu(), want, rtol=1e-5, atol=1e-5)
    with pytest.warns(ops.TorchRouteWarning, match="s_mul_e_sum"):
        got16 = ops.src_op_e_aggr_coo("mul", "sum", x.to(DEV).half(), ef.to(DEV).half(), row.to(DEV), col.to(DEV))
    assert got16.dtype == torch.float16 and torch.allclose(got16.float().cpu(), want, rtol=5e-2, atol=5e-2)
    with warnings.catch_warnings():  # the same (operator, reason) again: already noted
        warnings.simplefilter("error", ops.TorchRouteWarning)
        ops.src_op_e_aggr_coo("mul", "sum", x.to(DEV).half(), ef.to(DEV).half(), row.to(DEV), col.to(DEV))
    with pytest.warns(ops.TorchRouteWarning, match="scatter_add"):
        ops.scatter_add(ef.to(DEV).double(), row.to(DEV), n)


@pytest.mark.parametrize("k", [16, 64, 100])
@pytest.mark.parametrize("sort", [False, True], ids=["shuffled", "sorted"])
def test_xcd_plan_of_the_sorted_view_keeps_the_exact_rows(oracle, k, sort, monkeypatch):
    """cogdl_hip_gspmm_xcd (COGDL_AMD_XCD=force): the same sums over the length-ordered plan of the destination-sorted view --
    rows up to the exact-row bound bit-exact against the oracle (caller's edge order), hub rows up to re-association; forward
    AND the fused backward (the source-sorted view takes its own plan), scatter_add / op_aggr('mean') too."""
    from cogdl_amd import xcdplan

    monkeypatch.setattr(xcdplan, "MODE", "force")
    ops.clear_plans()
    n, e = 5000, 120000
    row, col = _coo(n, e, seed=11, sort=sort, hub=((3, 129), (4, 20000), (17, 3000), (18, 700), (4999, 2)))
    gen = torch.Generator().manual_seed(k)
    x, ef, w = torch.randn(n, k, generator=gen), torch.randn(e, k, generator=gen), torch.rand(e, generator=gen)
    g = _graph(row.to(DEV), col.to(DEV), w.to(DEV))
    thresh = _lib.hip().cogdl_hip_exact_row_edges(e)
    short = (torch.bincount(row, minlength=n) <= thresh).numpy()
    for op1, op2 in (("mul", "sum"), ("add", "mean"), ("sub", "sum")):
        xd, efd = x.to(DEV).requires_grad_(), ef.to(DEV).requires_grad_()
        got = getattr(ops, "s_%s_e_%s" % (op1, op2))(g, xd, efd, weight=True)
        plan = ops.edge_plan(g.edge_index[0], n)
        assert plan._xcd is not None, "the forced plan was not taken"
        want = oracle.src_op_e_aggr(op1, op2, x, ef, row, col, n, w=w)
        gn = got.detach().cpu().numpy()
        assert gn[short].tobytes() == want[short].tobytes()
        # hub rows: against float64 (the oracle's own fp32 sequential sum of 20,000 terms is the less accurate of the two)
        xs, es = x.double()[col], ef.double()
        msg = {"mul": xs * es, "add": xs + es, "sub": xs - es}[op1] * w.double().view(-1, 1)
        ref, scale = torch.zeros(n, k, dtype=torch.float64), torch.zeros(n, k, dtype=torch.float64)
        ref.index_add_(0, row, msg)
        scale.index_add_(0, row, msg.abs())
        if op2 == "mean":
            cnt = torch.bincount(row, minlength=n).clamp(min=1).double().view(-1, 1)
            ref, scale = ref / cnt, scale / cnt
        assert bool(((got.detach().cpu().double() - ref).abs() <= 1e-5 * scale + 1e-30).all())
        # backward: against the same operator with the plans off
        gout = torch.randn(n, k, generator=gen).to(DEV)
        got.backward(gout)
        gx, ge = xd.grad.clone(), efd.grad.clone()
        monkeypatch.setattr(xcdplan, "MODE", "off")
        xo, efo = x.to(DEV).requires_grad_(), ef.to(DEV).requires_grad_()
        getattr(ops, "s_%s_e_%s" % (op1, op2))(g, xo, efo, weight=True).backward(gout)
        monkeypatch.setattr(xcdplan, "MODE", "force")
        assert torch.equal(ge, efo.grad)  # (per-edge kernel: no plan involved)
        out_deg = torch.bincount(col, minlength=n)
        short_src = (out_deg <= thresh).to(DEV)
        assert torch.equal(gx[short_src], xo.grad[short_src])
        torch.testing.assert_close(gx, xo.grad, rtol=1e-4, atol=1e-4)
    data = torch.randn(e, k, generator=gen)
    for mean in (False, True):
        got = ops.op_aggr("mean" if mean else "sum", data.to(DEV), g.edge_index[0], n).cpu().numpy()
        want, scale = np.zeros((n, k), dtype=np.float64), np.zeros((n, k), dtype=np.float64)
        np.add.at(want, row.numpy(), data.numpy().astype(np.float64))
        np.add.at(scale, row.numpy(), np.abs(data.numpy()).astype(np.float64))
        if mean:
            cnt = np.maximum(np.bincount(row.numpy(), minlength=n), 1)[:, None]
            want, scale = want / cnt, scale / cnt
        assert np.all(np.abs(got - want) <= 1e-5 * scale + 1e-30)


def test_xcd_plan_is_taken_by_skewed_edge_lists_from_their_second_use_on(monkeypatch):
    from cogdl_amd import synth, xcdplan

    monkeypatch.setattr(xcdplan, "MODE", "auto")
    ops.clear_plans()
    for topo, expect in (("rmat", True), ("uniform", False)):
        gr = synth.arxiv_like(seed=0, topology=topo)
        deg = (gr.rowptr[1:] - gr.rowptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(gr.num_nodes), deg)
        shuffle = torch.randperm(row.numel(), generator=torch.Generator().manual_seed(1))
        row, col = row[shuffle].to(DEV), gr.colind.long()[shuffle].to(DEV)
        g = _graph(row, col, None)
        x, ef = torch.randn(gr.num_nodes, 96, device=DEV), torch.randn(row.numel(), 96, device=DEV)
        assert ops.s_mul_e_sum(g, x[:, :32].contiguous(), ef[:, :32].contiguous()).shape[1] == 32  # (narrow rows never take it)
        ops.clear_plans()
        first = ops.s_mul_e_sum(g, x, ef)
        plan = ops.edge_plan(row, gr.num_nodes)
        assert plan._xcd is None  # first use: the ordinary launch
        second = ops.s_mul_e_sum(g, x, ef)
        assert (plan._xcd is not None) == expect
        torch.testing.assert_close(second, first, rtol=1e-3, atol=1e-3)  # (hub rows: re-association under cancellation)
        ops.clear_plans()
