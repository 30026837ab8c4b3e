"""SURVEY.md section 8f rank 3 on the GPU: the fused normalisation / bias / ReLU epilogue of csr_spmm against the unfused
composition CogDL's dispatcher performs (cogdl/utils/spmm_utils.py:98-109), against the reference's own SAGELayer output
(golden, in_norm from Graph.row_norm()), its autograd against the oracle; and the hipGraph capture of a whole GCN
training step against the same step run eagerly."""
import numpy as np
import pytest
import torch

from cogdl_amd import _lib, graphs, synth
from cogdl_amd.operators.spmm import csr_spmm_epilogue_raw, csr_spmm_raw, csrspmm, csrspmm_fused

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("k", [1, 7, 40, 64, 128, 256])
@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("relu,bias", [(False, False), (True, True)])
def test_epilogue_is_bit_identical_to_the_dispatchers_composition(k, weighted, relu, bias):
    g = synth.random_csr(400, 300, 9, seed=k, weighted=weighted).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(k)
    x = torch.randn(300, k, device=DEV, generator=gen)
    out_norm = torch.rand(300, 1, device=DEV, generator=gen) + 0.1
    in_norm = torch.rand(400, 1, device=DEV, generator=gen) + 0.1
    b = torch.randn(k, device=DEV, generator=gen) if bias else None
    got = csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x, out_norm, in_norm, b, relu)
    want = in_norm * csr_spmm_raw(g.rowptr, g.colind, g.weight, out_norm * x)  # spmm_utils.py:99-109
    if bias:
        want = want + b
    if relu:
        want = torch.relu(want)
    assert torch.equal(got, want)
    # each scale on its own
    assert torch.equal(csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x, None, in_norm),
                       in_norm * csr_spmm_raw(g.rowptr, g.colind, g.weight, x))
    assert torch.equal(csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x, out_norm, None),
                       csr_spmm_raw(g.rowptr, g.colind, g.weight, out_norm * x))


def test_epilogue_hub_rows_and_full_size_graph(oracle):
    g = synth.arxiv_like(seed=0, topology="rmat")  # hub rows: the long-row path finishes rows in the combine kernel
    gd = g.to(DEV)
    n = g.num_nodes
    gen = torch.Generator().manual_seed(0)
    x, s_out, s_in = torch.randn(n, 64, generator=gen), torch.rand(n, generator=gen) + 0.5, torch.rand(n, generator=gen) + 0.5
    got = csr_spmm_epilogue_raw(gd.rowptr, gd.colind, gd.weight, x.to(DEV), s_out.to(DEV), s_in.to(DEV), None, True).cpu().numpy()
    ref = oracle.csr_spmm(g.rowptr, g.colind, g.weight, s_out.view(-1, 1) * x)
    want = np.maximum(s_in.view(-1, 1).numpy() * ref, 0)
    scale = s_in.view(-1, 1).numpy() * oracle.csr_spmm_abs(g.rowptr, g.colind, g.weight, s_out.view(-1, 1) * x)
    short = np.diff(g.rowptr.numpy()) <= _lib.hip().cogdl_hip_exact_row_edges(g.nnz)
    assert got[short].tobytes() == want[short].astype(np.float32).tobytes()
    assert np.all(np.abs(got - want) <= 1e-5 * scale + 1e-6)


def test_fused_mean_aggregation_reproduces_the_reference_sage_layer(golden):
    """in_norm = 1 / in-degree is what Graph.row_norm() leaves on a CSR-only block (cogdl/data/data.py:248-252);
    SAGELayer(mean) = fc(cat(x, spmm(block, x))) (layers/sage_layer.py:8-12,69-87)."""
    z = golden("sage_layer")
    rp, ci = T(z["block_row_indptr"]), T(z["block_col_indices"])
    deg = (rp[1:] - rp[:-1]).float()
    in_norm = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg))
    x = T(z["x_src"])
    agg = csrspmm_fused(rp.int(), ci.int(), x, None, None, in_norm)
    out = torch.nn.functional.linear(torch.cat([x, agg], dim=-1), T(z["fc_W"]), T(z["fc_b"]))
    np.testing.assert_allclose(out.cpu().numpy(), z["out"], rtol=1e-4, atol=1e-5)


def test_fused_autograd_equals_unfused_autograd_and_oracle(oracle):
    g = synth.random_csr(250, 180, 8, seed=3)
    gd = g.to(DEV)
    gen = torch.Generator().manual_seed(1)
    x, gout = torch.randn(180, 24, generator=gen), torch.randn(250, 24, generator=gen)
    s_out, s_in, b = torch.rand(180, 1, generator=gen) + 0.2, torch.rand(250, 1, generator=gen) + 0.2, torch.randn(24, generator=gen)
    xa, ba = x.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    ya = csrspmm_fused(gd.rowptr, gd.colind, xa, gd.weight, s_out.to(DEV), s_in.to(DEV), ba, True)
    ya.backward(gout.to(DEV))
    xb, bb = x.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yb = torch.relu(s_in.to(DEV) * csrspmm(gd.rowptr, gd.colind, s_out.to(DEV) * xb, gd.weight, False) + bb)
    yb.backward(gout.to(DEV))
    assert torch.equal(ya, yb)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), xb.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ba.grad.cpu().numpy(), bb.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # and against the oracle: grad_x = out_norm * (A^T (in_norm * (g * [y > 0])))
    gm = (gout * (ya.detach().cpu() > 0)) * s_in
    colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=180)
    want = s_out.numpy() * oracle.csr_spmm(colptr, rowind, w_t, gm)
    np.testing.assert_allclose(xa.grad.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_epilogue_declines_half_precision_and_bad_calls():
    g = synth.random_csr(20, 20, 3, seed=1).to(DEV)
    x = torch.randn(20, 8, device=DEV)
    with pytest.raises(_lib.BackendError):
        csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x.half())
    with pytest.raises(_lib.BackendError):
        csr_spmm_epilogue_raw(g.rowptr, g.colind, g.weight, x, torch.ones(19, device=DEV))


def _gcn_step_factory(seed, n=20000, f=32, hidden=16, classes=5):
    """CogDL's default gcn (2 x GCNLayer: spmm(graph, linear(x)), relu between) in capture-safe form: index tensor
    instead of a boolean mask, capturable Adam, no host synchronisation."""
    g = synth.scaled(n, 10, seed=seed, topology="rmat").to(DEV)
    rp64, ci64 = g.rowptr.long(), g.colind.long()
    torch.manual_seed(seed)
    lin1, lin2 = torch.nn.Linear(f, hidden).to(DEV), torch.nn.Linear(hidden, classes).to(DEV)
    x = torch.randn(n, f, device=DEV)
    y = torch.randint(0, classes, (n,), device=DEV)
    idx = torch.nonzero(torch.rand(n, device=DEV) < 0.5).flatten()
    y_idx = y[idx]
    params = list(lin1.parameters()) + list(lin2.parameters())
    opt = torch.optim.Adam(params, lr=0.01, capturable=True)
    loss_out = torch.zeros((), device=DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        h = torch.relu(csrspmm(rp64.int(), ci64.int(), lin1(x), g.weight, True))
        out = csrspmm(rp64.int(), ci64.int(), lin2(h), g.weight, True)
        loss = torch.nn.functional.cross_entropy(out.index_select(0, idx), y_idx)
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())
        return loss_out

    return step, params


@pytest.mark.parametrize("mfma_linear", [False, True], ids=["torch-linear", "mfma-linear"])
def test_hip_graph_capture_of_a_gcn_training_step_equals_eager(mfma_linear):
    """(mfma-linear: the split-K weight gradient zeroes its accumulator first -- by a kernel node, common.h
    fill_u32_async; a memset node that does not replay would make the replays accumulate into stale gradients.)"""
    from cogdl_amd import linear as cogdl_linear

    if mfma_linear:
        cogdl_linear.install()
    try:
        step_e, params_e = _gcn_step_factory(0)
        losses_e = [float(step_e()) for _ in range(3 + 4)]  # capture() runs 3 eager warm-up steps before replaying
        step_c, params_c = _gcn_step_factory(0)
        captured = graphs.capture(step_c, warmup=3)
        poison = torch.ones(32 << 20, dtype=torch.bool, device=DEV)
        del poison
        losses_c = [float(captured()) for _ in range(4)]
    finally:
        cogdl_linear.uninstall()
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=1e-5)
    for a, b in zip(params_c, params_e):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_captured_step_owns_its_plans_cache_eviction_and_new_weights_do_not_corrupt_replays():
    """Round-2 advisor finding: the captured kernels read the plans' colptr / rowind / perm and the memoised transposed
    weights by raw pointer; the plan cache used to be their only owner.  After the capture the cache is emptied, the
    plans' weight memo is replaced (another weight tensor on the same structure) and the freed memory is overwritten --
    the replays must still equal the eager steps."""
    import gc

    from cogdl_amd.plan import PLANS

    step_e, params_e = _gcn_step_factory(1)
    losses_e = [float(step_e()) for _ in range(3 + 3)]
    step_c, params_c = _gcn_step_factory(1)
    captured = graphs.capture(step_c, warmup=3)
    plans = list(PLANS.lru.values())
    assert plans
    for plan in plans:
        plan.transposed_values(torch.rand(plan.nnz, device=DEV))  # replaces plan._val_t / _val_src
    PLANS.clear()
    del plans, plan
    gc.collect()
    poison = [torch.full((48 << 20,), 0x7F, dtype=torch.uint8, device=DEV) for _ in range(4)]  # lands on whatever was freed
    torch.cuda.synchronize()
    del poison
    losses_c = [float(captured()) for _ in range(3)]
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=1e-5)
    for a, b in zip(params_c, params_e):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_captured_step_on_a_skewed_graph_runs_over_the_recorded_xcd_plans():
    """Round 6: the recorded eager run of graphs.capture waits for the structure's key, so every csr_spmm launch of the captured
    step -- forward calls too -- takes the length-ordered plan a known skewed structure takes (cogdl_amd/xcdplan.py:
    ordered_wanted; an eager forward call whose hash is in flight cannot).  The tape holds the decisions, the captured step
    owns the plans: the plan caches are emptied and the freed memory overwritten before the replays, which must equal the
    eager steps up to the re-association of the rows beyond the exact-row bound."""
    import gc

    from cogdl_amd import plan as plan_mod, xcdplan

    step_e, _ = _gcn_step_factory(2, n=60000)
    losses_e = [float(step_e()) for _ in range(3 + 3)]
    step_c, _ = _gcn_step_factory(2, n=60000)
    tapes, set_tape = [], plan_mod.set_tape

    def spy(tape):
        if tape is not None and tape not in tapes:
            tapes.append(tape)
        set_tape(tape)

    plan_mod.set_tape = spy
    try:
        captured = graphs.capture(step_c, warmup=3)
    finally:
        plan_mod.set_tape = set_tape
    (tape,) = tapes
    assert [k for k, _ in tape.choices] == ["csr_spmm.forward"] * 2 + ["csr_spmm.backward"] * 2
    for _, (split, xplan) in tape.choices:
        assert split is not None and isinstance(xplan, xcdplan.XcdPlan), "a skewed structure's launch recorded without a plan"
    assert tape.cpos == len(tape.choices) and tape.pos == len(tape.plans)
    xcdplan.XPLANS.clear()
    plan_mod.PLANS.clear()
    del tape, tapes
    gc.collect()
    poison = [torch.full((48 << 20,), 0x7F, dtype=torch.uint8, device=DEV) for _ in range(4)]
    torch.cuda.synchronize()
    del poison
    losses_c = [float(captured()) for _ in range(3)]
    np.testing.assert_allclose(losses_c, losses_e[3:], rtol=1e-4)
