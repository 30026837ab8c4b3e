"""BASELINE.json configs[2] at its TRUE size and dtype: GAT attention on the Reddit-shaped graph (N = 232,965,
114,848,857 nnz = Reddit's 114,615,892 directed edges + self loops, R-MAT rows up to ~10^5 edges), H = 8 heads x F = 8,
fp32 and bf16 -- every operator of the attention path against the CPU oracle (OpenMP, fp64 accumulation), computed once
per session (round-1 verdict, weak #1: this size used to be checked only against the HIP operators themselves).

Tolerances: an fp32 evaluation of a sum differs from the fp64 oracle by a multiple of eps * sum|terms|, so every check
is   |got - want| <= tol * (sum of |terms| of that output element)   with tol = 2e-5 for fp32 and 2^-7 for bf16 I/O
(inputs rounded to bf16 first; the kernels accumulate in fp32 and round once on store); the oracle provides the
sums of absolute terms.  Property checks (row sums, determinism, convexity) are kept as extras.
"""
import numpy as np
import pytest
import torch

from cogdl_amd import synth
from cogdl_amd.operators.edge_softmax import csr_edge_softmax
from cogdl_amd.operators.fused_gat import fused_gat_func, gat_forward
from cogdl_amd.operators.mhspmm import mhspmm_raw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, H, F = synth.REDDIT_NODES, 8, 8
TOL32, TOL16 = 2e-5, 2.0 ** -7


class Case:
    pass


@pytest.fixture(scope="module")
def reddit(oracle):
    c = Case()
    g = synth.reddit_like(seed=0, device=DEV)
    assert g.nnz == 2 * synth.REDDIT_UNDIRECTED + N == 114_848_857
    c.g = g
    c.rowptr, c.colind = g.rowptr.cpu().numpy(), g.colind.cpu().numpy()
    assert int(g.degrees().max()) > 50_000
    gen = torch.Generator(device=DEV).manual_seed(0)
    c.ar, c.ac = (torch.randn(N, H, device=DEV, generator=gen) for _ in range(2))
    c.feat = torch.randn(N, H, F, device=DEV, generator=gen)
    c.gout = torch.randn(N, H, F, device=DEV, generator=gen)
    return c


def _close(got, want, scale, tol, what):
    got, want, scale = (np.asarray(a, dtype=np.float64) for a in (got, want, scale))
    err = np.abs(got - want)
    bound = tol * scale + 1e-30
    worst = np.argmax(err / bound)
    assert np.all(err <= bound), "%s: err %.3e > bound %.3e at flat index %d (want %.6e)" % (
        what, err.flat[worst], bound.flat[worst], worst, want.flat[worst])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)], ids=["f32", "bf16"])
def test_fused_gat_forward_and_backward_vs_oracle(oracle, reddit, dtype, tol):
    c = reddit
    feat, gout = c.feat.to(dtype), c.gout.to(dtype)
    ar, ac, ft = c.ar.clone().requires_grad_(), c.ac.clone().requires_grad_(), feat.clone().requires_grad_()
    out = fused_gat_func(ar, ac, c.g.rowptr, c.g.colind, c.g.rowptr, c.g.colind, 0.2, ft)
    assert out.dtype == dtype
    out.backward(gout)
    torch.cuda.synchronize()
    h_l, h_r = c.ar.cpu().numpy(), c.ac.cpu().numpy()
    feat_h, gout_h = feat.float().cpu().numpy(), gout.float().cpu().numpy()
    want = oracle.gat_fwd(c.rowptr, c.colind, h_l, h_r, feat_h, 0.2)
    scale = oracle.gat_fwd(c.rowptr, c.colind, h_l, h_r, np.abs(feat_h), 0.2)  # sum_e a_e |feat_e|
    _close(out.detach().float().cpu().numpy(), want, scale, tol, "fused forward")
    assert float(out.detach().float().abs().max()) <= float(feat.float().abs().max()) * (1 + 2 * tol)  # convex combination
    gf, gl, gr, sf, sl, sr = oracle.gat_bwd(c.rowptr, c.colind, h_l, h_r, feat_h, 0.2, gout_h, scales=True)
    _close(ft.grad.float().cpu().numpy(), gf, sf, tol, "grad_feat")
    # the attention gradients are computed and returned in fp32 for every I/O dtype; their inputs (feat, grad_out and
    # -- through D = <g, out> -- the ROUNDED forward output) carry the dtype's rounding: same bound, the dtype's tol
    _close(ar.grad.cpu().numpy(), gl, sl, tol * (4 if dtype != torch.float32 else 1), "grad_attn_row")
    _close(ac.grad.cpu().numpy(), gr, sr, tol * (4 if dtype != torch.float32 else 1), "grad_attn_col")


def test_fused_gat_with_attention_dropout_vs_oracle_bf16(oracle, reddit):
    """The branch the gat model takes BY DEFAULT (attn_drop 0.5, cogdl/models/nn/gat.py:30; layers/gat_layer.py:72-77) at
    configs[2]'s true size and dtype: fused_gat_dropout_func in bf16 against the OpenMP oracle's fp64 composition WITH THE
    SAME MASK (the device mask, whose first two million edges are also checked against the oracle's Philox restatement).
    The layer's second shape (H = 1 x F = 41: rows padded to 48 inside the operator) rides along in the forward."""
    from cogdl_amd.operators.fused_gat import edge_dropout_mask, fused_gat_dropout_func

    c = reddit
    p, seed, tol = 0.5, 20260922, TOL16
    mask = edge_dropout_mask(c.g.nnz, H, p, seed, DEV)
    assert np.array_equal(mask[:2_000_000].cpu().numpy(), oracle.edge_dropout_mask(2_000_000, H, p, seed))
    kept = float((mask > 0).float().mean())
    assert abs(kept - 0.5) < 1e-3 and float(mask.max()) == 2.0
    feat, gout = c.feat.bfloat16(), c.gout.bfloat16()
    ar, ac, ft = c.ar.clone().requires_grad_(), c.ac.clone().requires_grad_(), feat.clone().requires_grad_()
    out = fused_gat_dropout_func(ar, ac, c.g.rowptr, c.g.colind, 0.2, ft, p, seed)
    assert out.dtype == torch.bfloat16
    out.backward(gout)
    torch.cuda.synchronize()
    drop = mask.cpu().numpy()
    del mask
    h_l, h_r = c.ar.cpu().numpy(), c.ac.cpu().numpy()
    feat_h, gout_h = feat.float().cpu().numpy(), gout.float().cpu().numpy()
    want = oracle.gat_fwd(c.rowptr, c.colind, h_l, h_r, feat_h, 0.2, drop=drop)
    scale = oracle.gat_fwd(c.rowptr, c.colind, h_l, h_r, np.abs(feat_h), 0.2, drop=drop)
    _close(out.detach().float().cpu().numpy(), want, scale, tol, "fused dropout forward")
    gf, gl, gr, sf, sl, sr = oracle.gat_bwd(c.rowptr, c.colind, h_l, h_r, feat_h, 0.2, gout_h, scales=True, drop=drop)
    _close(ft.grad.float().cpu().numpy(), gf, sf, tol, "dropout grad_feat")
    _close(ar.grad.cpu().numpy(), gl, sl, 4 * tol, "dropout grad_attn_row")
    _close(ac.grad.cpu().numpy(), gr, sr, 4 * tol, "dropout grad_attn_col")
    # the second layer's shape, forward only: one head, 41 features (odd: 82-byte rows, padded to 48 by the operator)
    gen = torch.Generator(device=DEV).manual_seed(41)
    ar1, ac1 = torch.randn(N, 1, device=DEV, generator=gen), torch.randn(N, 1, device=DEV, generator=gen)
    f41 = torch.randn(N, 1, 41, device=DEV, generator=gen).bfloat16()
    out41 = fused_gat_dropout_func(ar1, ac1, c.g.rowptr, c.g.colind, 0.2, f41, p, seed + 1)
    drop1 = edge_dropout_mask(c.g.nnz, 1, p, seed + 1, DEV).cpu().numpy()
    f41_h = f41.float().cpu().numpy()
    want41 = oracle.gat_fwd(c.rowptr, c.colind, ar1.cpu().numpy(), ac1.cpu().numpy(), f41_h, 0.2, drop=drop1)
    scale41 = oracle.gat_fwd(c.rowptr, c.colind, ar1.cpu().numpy(), ac1.cpu().numpy(), np.abs(f41_h), 0.2, drop=drop1)
    _close(out41.float().cpu().numpy(), want41, scale41, tol, "fused dropout forward H=1 F=41")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOL16)], ids=["f32", "bf16"])
@pytest.mark.parametrize("scale_in", [1.0, 10.0], ids=["N(0,1)", "x10"])
def test_edge_softmax_forward_backward_vs_oracle(oracle, reddit, dtype, tol, scale_in):
    c = reddit
    gen = torch.Generator(device=DEV).manual_seed(int(scale_in))
    score = (torch.randn(c.g.nnz, H, device=DEV, generator=gen) * scale_in).to(dtype)
    grad = torch.randn(c.g.nnz, H, device=DEV, generator=gen).to(dtype)
    sd = score.clone().requires_grad_()
    att = csr_edge_softmax(c.g.rowptr, sd)
    assert att.dtype == dtype
    att.backward(grad)
    again = csr_edge_softmax(c.g.rowptr, score)
    assert torch.equal(again, att.detach())  # run-to-run identical: fixed merge order of the tile records
    from cogdl_amd.operators import edge_softmax as es_mod
    assert int(es_mod.LAST_WORKSPACE[:4].view(torch.int32)[0]) == 0, "a cross-tile wait timed out (escape path taken)"
    want = oracle.edge_softmax_fwd(c.rowptr, score.float().cpu().numpy())
    got = att.detach().float().cpu().numpy()
    _close(got, want, want, tol, "edge_softmax forward")  # every term is positive: the scale is the value itself
    row = torch.repeat_interleave(torch.arange(N, device=DEV), c.g.degrees().to(DEV))
    sums = torch.zeros(N, H, device=DEV).index_add_(0, row, att.detach().float())
    assert torch.allclose(sums, torch.ones_like(sums), atol=(2e-4 if dtype == torch.float32 else 2e-2))
    # backward: g_in = s * (g - sum_row s*g), from the SAME (rounded) softmax the kernel was given
    sm_h, gr_h = got, grad.float().cpu().numpy()
    want_g = oracle.edge_softmax_bwd(c.rowptr, sm_h, gr_h)
    dot_abs = torch.zeros(N, H, device=DEV).index_add_(0, row, (att.detach().float() * grad.float()).abs())
    scale_g = (att.detach().float() * (grad.float().abs() + dot_abs[row])).cpu().numpy()
    _close(sd.grad.float().cpu().numpy(), want_g, scale_g, tol, "edge_softmax backward")


def test_mhspmm_and_unfused_composition_vs_oracle(oracle, reddit):
    c = reddit
    gen = torch.Generator(device=DEV).manual_seed(5)
    att = csr_edge_softmax(c.g.rowptr, torch.randn(c.g.nnz, H, device=DEV, generator=gen))
    out = mhspmm_raw(c.g.rowptr, c.g.colind, att, c.feat)
    att_h, feat_h = att.cpu().numpy(), c.feat.cpu().numpy()
    want = oracle.mhspmm(c.rowptr, c.colind, att_h, feat_h)
    scale = oracle.mhspmm(c.rowptr, c.colind, att_h, np.abs(feat_h))
    _close(out.cpu().numpy(), want, scale, 1e-5, "mhspmm")
    # fused == unfused composition on the GPU as well (what GATLayer's two branches promise each other)
    row = torch.repeat_interleave(torch.arange(N, device=DEV), c.g.degrees().to(DEV))
    score = torch.nn.functional.leaky_relu(c.ar[row] + c.ac[c.g.colind.long()], 0.2)
    unfused = mhspmm_raw(c.g.rowptr, c.g.colind, csr_edge_softmax(c.g.rowptr, score), c.feat)
    fused, emax, _ = gat_forward(c.ar, c.ac, c.g.rowptr, c.g.colind, 0.2, c.feat)
    assert torch.allclose(fused, unfused, rtol=1e-4, atol=1e-4)
    mx = torch.full((N, H), -float("inf"), device=DEV).scatter_reduce(0, row.view(-1, 1).expand_as(score), score, "amax")
    assert torch.allclose(emax, mx, rtol=0, atol=1e-6)
