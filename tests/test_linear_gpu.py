"""linear_wgrad (the dense side of a layer): grad_w = grad_out^T . x and grad_b on tall-skinny
operands through v_mfma_f32_32x32x2_f32, against float64 torch on the same inputs (tolerance 1e-5 of the sum of the
magnitudes of the terms -- fp32 accumulation over up to 1.7e5 rows), and the torch.nn.functional.linear hook."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("k,in_f,out_f", [(1000, 128, 64), (169_343, 128, 64), (169_343, 64, 40), (5003, 100, 47),
                                          (777, 64, 40), (4097, 602, 8), (300, 33, 7), (2000, 512, 256), (9, 4, 4),
                                          (1, 128, 64), (65, 16, 1), (12345, 1, 5)])
def test_wgrad_matches_float64(k, in_f, out_f):
    from cogdl_amd.linear import linear_wgrad

    gen = torch.Generator().manual_seed(k + in_f)
    x = torch.randn(k, in_f, generator=gen)
    g = torch.randn(k, out_f, generator=gen)
    # asymmetric structure: catches row/column or tile-permutation mix-ups that random data of equal scale would too,
    # but with an exact expected pattern
    x[:, 0] += 3.0
    g[:, -1] -= 2.0
    gw, gb = linear_wgrad(x.to(DEV), g.to(DEV))
    want_w = (g.double().t() @ x.double()).numpy()
    want_b = g.double().sum(0).numpy()
    scale_w = (g.abs().double().t() @ x.abs().double()).numpy()
    assert gw.shape == (out_f, in_f) and gb.shape == (out_f,)
    assert np.all(np.abs(gw.cpu().numpy() - want_w) <= 1e-5 * scale_w + 1e-6)
    assert np.all(np.abs(gb.cpu().numpy() - want_b) <= 1e-5 * g.abs().double().sum(0).numpy() + 1e-6)
    again, _ = linear_wgrad(x.to(DEV), g.to(DEV), want_bias=False)
    assert torch.equal(again, gw)  # fixed reduction order: run-to-run identical


def test_one_hot_rows_give_exact_placement():
    """x = one-hot rows, grad_out = distinct integers: every output element is a single exactly representable sum."""
    from cogdl_amd.linear import linear_wgrad

    k, in_f, out_f = 4096, 128, 64
    cols = torch.arange(k) % in_f
    x = torch.zeros(k, in_f)
    x[torch.arange(k), cols] = 1.0
    g = (torch.arange(k * out_f, dtype=torch.float32).view(k, out_f) % 251) - 125.0
    gw, gb = linear_wgrad(x.to(DEV), g.to(DEV))
    want = g.double().t() @ x.double()
    assert torch.equal(gw.cpu().double(), want)
    assert torch.equal(gb.cpu().double(), g.double().sum(0))


def test_functional_linear_hook_matches_torch():
    from cogdl_amd import linear as cl

    torch.manual_seed(0)
    lin = torch.nn.Linear(128, 64).to(DEV)
    x = torch.randn(20000, 128, device=DEV, requires_grad=True)
    gout = torch.randn(20000, 64, device=DEV)
    lin(x).backward(gout)
    ref = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    cl.install()
    try:
        y = lin(x)
        assert y.grad_fn is not None and "LinearFunction" in type(y.grad_fn).__name__
        y.backward(gout)
        small = torch.nn.functional.linear(torch.randn(8, 128, device=DEV), lin.weight, lin.bias)  # not covered: torch's own
        assert "LinearFunction" not in type(small.grad_fn).__name__
    finally:
        cl.uninstall()
    assert torch.allclose(x.grad, ref[0], rtol=1e-4, atol=1e-4)
    assert torch.allclose(lin.weight.grad, ref[1], rtol=1e-4, atol=2e-3)
    assert torch.allclose(lin.bias.grad, ref[2], rtol=1e-4, atol=2e-3)
    assert torch.nn.functional.linear is cl._orig_linear


@pytest.mark.parametrize("rows,k,n", [(1000, 128, 64), (169_343, 128, 64), (169_343, 64, 40), (5003, 100, 47), (33, 7, 3),
                                      (4097, 66, 33), (2000, 256, 64), (31, 300, 17), (1, 5, 1)])
@pytest.mark.parametrize("transposed_w", [True, False])
def test_tall_skinny_matmul_matches_float64(rows, k, n, transposed_w):
    from cogdl_amd.linear import tall_skinny_matmul

    gen = torch.Generator().manual_seed(rows + k + n)
    x = torch.randn(rows, k, generator=gen)
    w = torch.randn(n, k, generator=gen) if transposed_w else torch.randn(k, n, generator=gen)
    bias = torch.randn(n, generator=gen) if transposed_w else None
    x[:, 0] += 2.0  # asymmetric: row/column mix-ups show
    got = tall_skinny_matmul(x.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV), transposed_w)
    assert got is not None
    b64 = w.double().t() if transposed_w else w.double()
    want = x.double() @ b64 + (bias.double() if bias is not None else 0.0)
    scale = x.abs().double() @ b64.abs() + 1.0
    assert np.all(np.abs(got.cpu().numpy() - want.numpy()) <= 1e-5 * scale.numpy())


def test_tall_skinny_matmul_declines_big_weights():
    from cogdl_amd.linear import tall_skinny_matmul

    x = torch.randn(100, 1024, device=DEV)
    assert tall_skinny_matmul(x, torch.randn(64, 1024, device=DEV), None, True) is None   # 1024 x 64 x 4 B > 96 KB
    assert tall_skinny_matmul(x, torch.randn(65, 1024, device=DEV), None, True) is None   # > 64 columns
