"""The attention dropout of the fused GAT operator -- the branch CogDL's gat model takes by default (attn_drop 0.5:
cogdl/models/nn/gat.py:30; leaky_relu(h_l[row] + h_r[col]) -> edge_softmax -> nn.Dropout -> mhspmm,
cogdl/layers/gat_layer.py:72-77) -- as cogdl_hip_gat_dropout_fwd / _bwd.

CPU part (no GPU): the mask generator is the published Philox4x32-10; the oracle's numpy restatement is pinned against
Random123's known-answer vectors, and the library's own header code (run on the host) equals the oracle bit for bit.
GPU part: the device mask equals the oracle's; the fused forward / backward equal the oracle's fp64 composition with the
SAME mask (every H x F geometry, both forward kernels, hub rows, bf16), the keep rate passes a chi-square test and p = 0
is the plain operator."""
import numpy as np
import pytest
import torch

from cogdl_amd import _lib, synth

DEV = "cuda:0"


def rand(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# ------------------------------------------------------------------------------------------------ CPU: the generator
def test_philox_known_answers(oracle):
    """Random123 kat_vectors, philox4x32 with 10 rounds (Salmon et al., SC'11)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, want in kat:
        got = oracle.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(x) for x in got) == want


@pytest.mark.parametrize("n,h,p,seed", [(1000, 8, 0.5, 1), (777, 1, 0.6, 2 ** 40 + 5), (500, 12, 0.1, 99),
                                        (300, 64, 0.9, 7), (10, 3, 0.0, 1), (10, 3, 1.0, 1), (0, 4, 0.5, 3)])
def test_library_mask_on_the_host_equals_the_oracle(oracle, n, h, p, seed):
    m = np.full((n, h), -1.0, np.float32)
    rc = _lib.hip().cogdl_hip_edge_dropout_mask_host(n, h, p, seed, m.ctypes.data if n else None)
    assert rc == 0
    assert np.array_equal(m, oracle.edge_dropout_mask(n, h, p, seed))
    if 0 < p < 1 and n:
        assert set(np.unique(m)) <= {np.float32(0.0), m.max()} and abs(m.max() - 1 / (1 - p)) < 1e-3


def test_mask_statistics_on_the_host(oracle):
    """Keep rate (chi-square, 1 degree of freedom per head), unbiasedness E[d] = 1, independence of heads and of
    consecutive edges (correlations of 2*10^5 samples)."""
    n, h, p = 200_000, 8, 0.5
    m = oracle.edge_dropout_mask(n, h, p, 20260922) > 0
    kept = m.sum(axis=0)
    chi2 = (kept - n * (1 - p)) ** 2 / (n * p * (1 - p))
    assert chi2.max() < 15.1, chi2  # P[chi2_1 > 15.1] = 1e-4
    c = np.corrcoef(m.T.astype(np.float64))
    assert np.abs(c - np.eye(h)).max() < 0.01
    assert abs(np.corrcoef(m[:-1, 0], m[1:, 0])[0, 1]) < 0.01
    m2 = oracle.edge_dropout_mask(n, h, p, 20260923) > 0
    assert abs(np.corrcoef(m[:, 0], m2[:, 0])[0, 1]) < 0.01  # another seed, another mask
    d = oracle.edge_dropout_mask(n, 4, 0.3, 5)
    assert abs(d.mean() - 1.0) < 0.01


def test_invalid_arguments():
    lib = _lib.hip()
    m = np.zeros((4, 4), np.float32)
    assert lib.cogdl_hip_edge_dropout_mask_host(4, 4, 1.5, 0, m.ctypes.data) == 1
    assert lib.cogdl_hip_edge_dropout_mask_host(4, 4, float("nan"), 0, m.ctypes.data) == 1
    assert lib.cogdl_hip_edge_dropout_mask_host(4, 65, 0.5, 0, m.ctypes.data) == _lib.EUNSUPPORTED
    assert lib.cogdl_hip_edge_dropout_mask_host(4, 4, 0.5, 0, None) == 1


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n,h,p,seed", [(100_003, 8, 0.5, 11), (5000, 1, 0.25, 2 ** 50 + 1), (3000, 20, 0.7, 3),
                                        (1000, 64, 0.5, 4)])
def test_device_mask_equals_the_oracle(oracle, n, h, p, seed):
    from cogdl_amd.operators.fused_gat import edge_dropout_mask

    assert np.array_equal(edge_dropout_mask(n, h, p, seed, DEV).cpu().numpy(), oracle.edge_dropout_mask(n, h, p, seed))


@pytest.fixture(params=[0, 1, 2], ids=["auto", "edgewise-softmax", "chunkwise-softmax"])
def gat_kernel(request):
    _lib.hip().cogdl_hip_set_tuning(5, request.param)
    yield request.param
    _lib.hip().cogdl_hip_set_tuning(5, 0)


def _check(oracle, g, n_src, h, f, p, seed, dtype=torch.float32, rtol=2e-5, rtol_g=2e-5):
    from cogdl_amd.operators.fused_gat import fused_gat_dropout_func

    v = g.num_nodes
    a_row, a_col = rand(v, h, seed=seed), rand(n_src, h, seed=seed + 1)
    feat, gout = rand(n_src, h, f, seed=seed + 2).to(dtype), rand(v, h, f, seed=seed + 3).to(dtype)
    drop = oracle.edge_dropout_mask(g.nnz, h, p, seed)
    want = oracle.gat_fwd(g.rowptr, g.colind, a_row, a_col, feat.float(), 0.2, drop=drop)
    w_feat, w_l, w_r, s_feat, s_l, s_r = oracle.gat_bwd(g.rowptr, g.colind, a_row, a_col, feat.float(), 0.2,
                                                        gout.float(), n_src=n_src, scales=True, drop=drop)
    ar, ac, ft = (t.to(DEV).requires_grad_() for t in (a_row, a_col, feat))
    out = fused_gat_dropout_func(ar, ac, g.rowptr.to(DEV), g.colind.to(DEV), 0.2, ft, p, seed)
    assert out.dtype == dtype
    abs_out = oracle.gat_fwd(g.rowptr, g.colind, a_row, a_col, feat.float().abs(), 0.2, drop=drop)
    err = np.abs(out.detach().float().cpu().numpy() - want)
    assert np.all(err <= rtol * abs_out + 1e-30), err.max()
    out.backward(gout.to(DEV))
    for got, ref, scale, name in ((ft.grad, w_feat, s_feat, "feat"), (ar.grad, w_l, s_l, "attn_row"),
                                  (ac.grad, w_r, s_r, "attn_col")):
        e = np.abs(got.float().cpu().numpy() - ref)
        assert np.all(e <= rtol_g * scale + 1e-30), (name, float((e / (scale + 1e-30)).max()))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("h,f", [(8, 8), (4, 8), (1, 41), (2, 16), (1, 64), (4, 16), (3, 5), (8, 32), (6, 12), (8, 64),
                                 (4, 128), (16, 64), (33, 4), (64, 2)])
@pytest.mark.parametrize("pad", [True, False], ids=["padded-rows", "raw-width"])
def test_fused_gat_dropout_forward_backward_vs_oracle_with_the_same_mask(oracle, gat_kernel, h, f, pad, monkeypatch):
    """Tolerance: 2e-5 x the sum of the absolute values of the terms of each output (fp32 accumulation of fp32 inputs;
    the oracle supplies the sums)."""
    from cogdl_amd.operators import fused_gat

    monkeypatch.setattr(fused_gat, "PAD_FEATURES", pad)
    g = synth.random_csr(150, 120, 7, seed=h * 100 + f, weighted=False)
    _check(oracle, g, 120, h, f, 0.5, seed=1000 + h * f)


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.1, 0.6, 0.9])
def test_fused_gat_dropout_other_probabilities(oracle, p):
    g = synth.random_csr(300, 300, 9, seed=4, weighted=False)
    _check(oracle, g, 300, 8, 8, p, seed=int(p * 100))


HUBS = [((3, 129), (4, 1000), (17, 5000), (18, 257), (40, 128)), ((0, 4000),), ((59, 3000), (58, 131))]


@pytest.mark.gpu
@pytest.mark.parametrize("hubs", HUBS)
@pytest.mark.parametrize("h,f", [(8, 8), (1, 41), (8, 64), (6, 12)])
def test_fused_gat_dropout_hub_rows(oracle, gat_kernel, hubs, h, f):
    """Rows and columns of thousands of edges: the chunk-parallel long-row path regenerates the same mask piece by piece
    (forward and row pass by CSR position, column pass through the plan's permutation)."""
    g = synth.hub_csr(60, 60, hubs=hubs, seed=h * f, weighted=False)
    _check(oracle, g, 60, h, f, 0.5, seed=7 + h, rtol=4e-5, rtol_g=4e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h,f", [(8, 8), (1, 41), (3, 5)])
def test_fused_gat_dropout_16bit(oracle, dtype, h, f):
    """configs[2]'s dtype: features / outputs / their gradients in bf16 (read natively, fp32 arithmetic, one rounding on
    store): 2^-7 (bf16) / 2^-10 (f16) x the sum of absolute terms."""
    g = synth.random_csr(200, 200, 9, seed=3, weighted=False)
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    _check(oracle, g, 200, h, f, 0.5, seed=21, dtype=dtype, rtol=tol, rtol_g=tol)


@pytest.mark.gpu
def test_p_zero_is_the_plain_operator_and_seeds_differ():
    from cogdl_amd.operators.fused_gat import fused_gat_dropout_func, fused_gat_func

    g = synth.random_csr(500, 500, 12, seed=8, weighted=False)
    rp, ci = g.rowptr.to(DEV), g.colind.to(DEV)
    ar, ac, ft = rand(500, 8, seed=1).to(DEV), rand(500, 8, seed=2).to(DEV), rand(500, 8, 8, seed=3).to(DEV)
    plain = fused_gat_func(ar, ac, rp, ci, rp, ci, 0.2, ft)
    assert torch.equal(fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.0), plain)
    a = fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5, seed=1)
    assert torch.equal(a, fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5, seed=1))  # deterministic
    assert not torch.equal(a, fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5, seed=2))
    torch.manual_seed(5)
    b = fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5)  # seed drawn from torch's generator
    torch.manual_seed(5)
    assert torch.equal(b, fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5))
    # unbiased: the mean over many masks approaches the plain output
    acc = torch.zeros_like(plain)
    for s in range(64):
        acc += fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 0.5, seed=100 + s)
    assert float((acc / 64 - plain).abs().mean()) < 0.25 * float(plain.abs().mean())
    with pytest.raises(ValueError):
        fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, 1.5)


@pytest.mark.gpu
def test_fused_dropout_equals_the_unfused_hip_operators_with_the_exported_mask():
    """leaky_relu(h_l[row] + h_r[col]) -> csr_edge_softmax -> (x mask) -> csrmhspmm on the GPU, gradients by autograd
    through those operators, against the one fused operator."""
    from cogdl_amd.operators.edge_softmax import csr_edge_softmax
    from cogdl_amd.operators.fused_gat import edge_dropout_mask, fused_gat_dropout_func
    from cogdl_amd.operators.mhspmm import csrmhspmm

    g = synth.scaled(3000, 12, seed=2, norm=None)
    h, f, p, seed = 8, 8, 0.5, 424242
    rp, ci = g.rowptr.to(DEV), g.colind.to(DEV)
    row = torch.repeat_interleave(torch.arange(3000), g.degrees()).to(DEV)
    base = [rand(3000, h, seed=5), rand(3000, h, seed=6), rand(3000, h, f, seed=7)]
    gout = rand(3000, h, f, seed=8).to(DEV)
    ar, ac, ft = (t.to(DEV).requires_grad_() for t in base)
    fused = fused_gat_dropout_func(ar, ac, rp, ci, 0.2, ft, p, seed)
    fused.backward(gout)
    ar2, ac2, ft2 = (t.to(DEV).requires_grad_() for t in base)
    att = csr_edge_softmax(rp, torch.nn.functional.leaky_relu(ar2[row] + ac2[ci.long()], 0.2))
    unfused = csrmhspmm(rp, ci, ft2, att * edge_dropout_mask(g.nnz, h, p, seed, DEV))
    unfused.backward(gout)
    assert torch.allclose(fused, unfused, rtol=2e-5, atol=2e-6)
    for a, b in ((ar.grad, ar2.grad), (ac.grad, ac2.grad), (ft.grad, ft2.grad)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4)
