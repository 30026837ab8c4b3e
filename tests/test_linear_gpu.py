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


# ---- bf16 autocast product (csrc/linear_fwd16.hip, ABI v9) ---------------------------------------------------------------
def _bf16_round(t, half=torch.bfloat16):
    return t.to(half).double()


@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("x_dtype", [torch.float32, "half"], ids=["x-f32", "x-16bit"])
@pytest.mark.parametrize("rows,k,n,transposed_w", [
    (5000, 602, 64, False),   # configs[2], first layer: 2408-byte fp32 rows (8-byte loads), ragged last macro-step
    (5000, 64, 41, False),    # its second layer
    (4097, 128, 64, True), (1000, 33, 7, False), (1000, 40, 32, True), (33, 34, 1, False), (31, 16, 64, False),
    (1, 8, 3, True), (300, 1000, 33, False), (2049, 96, 48, True), (64, 2, 2, False)])
def test_bf16_matmul_matches_the_rounded_operands_in_float64(rows, k, n, transposed_w, x_dtype, half):
    """out = bf16( sum_k bf16(x)[r, k] * bf16(B)[k, c] + bias ): fp32 accumulation of exact bf16 x bf16 products, so the
    float64 product of the ROUNDED operands is matched to fp32-accumulation accuracy before the final rounding -- checked
    as: |got - want| <= one bf16 ulp of want + 2e-6 of the sum of magnitudes.  One-hot / asymmetric columns catch operand
    permutations (k order inside a macro-step, C/D register map)."""
    from cogdl_amd.linear import tall_skinny_matmul_16

    x_dtype = half if x_dtype == "half" else x_dtype
    ulp = 2.0 ** -8 if half == torch.bfloat16 else 2.0 ** -10  # (f16: 11 significant bits; one ulp of slack)
    gen = torch.Generator().manual_seed(rows + 7 * k + n)
    x = torch.randn(rows, k, generator=gen)
    w = torch.randn((n, k) if transposed_w else (k, n), generator=gen)
    x[:, 0] += 3.0
    x[:, -1] -= 1.5
    bias = torch.randn(n, generator=gen)
    if x_dtype != torch.float32 and k % 2:
        got = tall_skinny_matmul_16(x.to(DEV).to(x_dtype), w.to(DEV), bias.to(DEV), transposed_w, half)
        assert got is None  # (16-bit rows of odd length: declined, the caller keeps torch's product)
        return
    xd = x.to(DEV).to(x_dtype)
    got = tall_skinny_matmul_16(xd, w.to(DEV), bias.to(DEV), transposed_w, half)
    assert got is not None and got.dtype == half and got.shape == (rows, n)
    b = _bf16_round(w, half).t() if transposed_w else _bf16_round(w, half)
    want = _bf16_round(x, half) @ b + bias.double()
    scale = _bf16_round(x, half).abs() @ b.abs() + bias.double().abs()
    err = (got.cpu().double() - want).abs()
    bound = want.abs() * ulp + 2e-6 * scale + 1e-7  # (+ f16's subnormal spacing near zero)
    assert bool((err <= bound).all()), "max err/bound %.3f" % float((err / bound).max())
    # weights already in the 16-bit type (autocast's cached cast): same result
    again = tall_skinny_matmul_16(xd, w.to(DEV).to(half), bias.to(DEV), transposed_w, half)
    assert torch.equal(again, got)


def test_bf16_matmul_places_every_element_exactly():
    """x = one-hot rows (column r % K), W = distinct small integers: out[r, c] = W[r % K, c] exactly -- any mix-up of the
    k order inside a macro-step, of the half-waves or of the C/D register map shows as a wrong integer."""
    from cogdl_amd.linear import tall_skinny_matmul_bf16

    for rows, k, n in ((4096, 602, 64), (1000, 70, 33), (999, 32, 32)):
        cols = torch.arange(rows) % k
        x = torch.zeros(rows, k)
        x[torch.arange(rows), cols] = 1.0
        w = ((torch.arange(k * n, dtype=torch.float32).view(k, n) * 7) % 251) - 125.0  # exactly representable in bf16
        got = tall_skinny_matmul_bf16(x.to(DEV), w.to(DEV), None, False)
        assert torch.equal(got.cpu().float(), w[cols])


def test_bf16_matmul_does_not_leak_a_neighbouring_row_through_the_ragged_tail():
    """K = 602: the last macro-step of a row reads 22 elements of the NEXT row (B is zero there, but 0 * NaN is NaN)."""
    from cogdl_amd.linear import tall_skinny_matmul_bf16

    rows, k, n = 200, 602, 64
    x = torch.randn(rows, k)
    x[101, :30] = float("nan")
    x[150, 0] = float("inf")
    got = tall_skinny_matmul_bf16(x.to(DEV), torch.randn(k, n).to(DEV), None, False).cpu().float()
    bad = ~torch.isfinite(got).all(1)
    assert bad.nonzero().flatten().tolist() == [101, 150]


def test_matmul_under_bf16_autocast_matches_torch_autocast():
    """cogdl_amd.linear.matmul vs torch.matmul under the same autocast context: forward within a bf16 ulp, grad_W against
    the float64 product of the operands torch's autocast backward sees (bf16 x, bf16 grad)."""
    from cogdl_amd import linear as cl

    torch.manual_seed(1)
    rows, k, n = 20000, 602, 64
    x = torch.randn(rows, k, device=DEV)
    w0 = (torch.randn(k, n, device=DEV) * 0.05)
    gout = torch.randn(rows, n, device=DEV).bfloat16()
    outs = []
    for fn in (torch.matmul, cl.matmul):
        w = w0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(x, w)
        assert y.dtype == torch.bfloat16
        y.backward(gout)
        outs.append((y.detach().float(), w.grad.clone()))
    (y_t, gw_t), (y_c, gw_c) = outs
    assert gw_c.dtype == torch.float32 and gw_c.shape == (k, n)
    assert bool(((y_c - y_t).abs() <= y_t.abs() * 2.0 ** -7 + 1e-3).all())
    want = x.double().t() @ gout.double()
    scale = x.double().abs().t() @ gout.double().abs()
    assert bool(((gw_c.double() - want).abs() <= 1e-5 * scale + 1e-6).all())       # fp32 reduction of un-rounded x
    assert bool(((gw_t.double() - want).abs() <= 2.0 ** -7 * scale + 1e-6).all())  # (torch's own: bf16 operands, bf16 result)
    # outside autocast, or for shapes it does not cover, it IS torch.matmul
    w = w0.clone().requires_grad_()
    assert torch.equal(cl.matmul(x, w), torch.matmul(x, w))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not cl.matmul_covers(x[:100], w) and not cl.matmul_covers(x, torch.randn(k, 128, device=DEV))
    # second-layer shape: bf16 x that requires grad
    xb = torch.randn(rows, 64, device=DEV).bfloat16().requires_grad_()
    w2 = (torch.randn(64, 41, device=DEV) * 0.1).requires_grad_()
    g2 = torch.randn(rows, 41, device=DEV).bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = cl.matmul(xb, w2)
    y2.backward(g2)
    want_x = g2.double() @ w2.detach().bfloat16().double().t()
    assert xb.grad.dtype == torch.bfloat16 and bool(((xb.grad.double() - want_x).abs() <= want_x.abs() * 2.0 ** -7 + 1e-2).all())
    want_w = xb.detach().double().t() @ g2.double()
    assert bool(((w2.grad.double() - want_w).abs() <= 1e-5 * (xb.detach().double().abs().t() @ g2.double().abs()) + 1e-6).all())


def test_functional_linear_hook_under_bf16_autocast():
    """install(): nn.Linear under torch.autocast(bfloat16) on a tall input takes LinearBf16Function -- forward within a bf16
    ulp of torch's autocast product, grad_W / grad_b against float64 (fp32 reductions of the un-rounded input), grad_x against
    the bf16 product torch forms; shapes outside the cover (wide outputs, few rows) stay torch's."""
    from cogdl_amd import linear as cl

    torch.manual_seed(2)
    rows, k, n = 30000, 100, 47
    lin = torch.nn.Linear(k, n).to(DEV)
    x = torch.randn(rows, k, device=DEV, requires_grad=True)
    gout = torch.randn(rows, n, device=DEV).bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_t = lin(x)
    y_t.backward(gout)
    ref = (y_t.detach().float(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = lin.weight.grad = lin.bias.grad = None
    cl.install()
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert cl.covers_bf16(x, lin.weight, lin.bias)
            y = lin(x)
            wide = torch.nn.Linear(k, 128).to(DEV)
            assert not cl.covers_bf16(x, wide.weight, wide.bias) and not cl.covers_bf16(x[:100], lin.weight, lin.bias)
            assert wide(x).dtype == torch.bfloat16
        assert y.dtype == torch.bfloat16
        y.backward(gout)
    finally:
        cl.uninstall()
    assert bool(((y.detach().float() - ref[0]).abs() <= ref[0].abs() * 2.0 ** -7 + 2e-3).all())
    g64, x64 = gout.double(), x.detach().double()
    want_w, scale_w = g64.t() @ x64, g64.abs().t() @ x64.abs()
    assert lin.weight.grad.dtype == torch.float32 and bool(((lin.weight.grad.double() - want_w).abs() <= 1e-5 * scale_w + 1e-6).all())
    assert bool(((lin.bias.grad.double() - g64.sum(0)).abs() <= 1e-5 * g64.abs().sum(0) + 1e-6).all())
    assert x.grad.dtype == torch.float32 and bool(((x.grad - ref[1]).abs() <= ref[1].abs() * 2.0 ** -6 + 2e-2).all())


def test_functional_linear_hook_under_fp16_autocast_with_a_grad_scaler():
    """The reference's own mixed precision (Trainer(fp16=True): torch.cuda.amp.autocast + GradScaler, cogdl/trainer/trainer.py):
    nn.Linear on a tall input under float16 autocast takes cogdl_hip_linear_fwd_f16; the scaled gradients come out as torch's."""
    from cogdl_amd import linear as cl

    torch.manual_seed(3)
    rows, k, n = 20000, 128, 40
    x = torch.randn(rows, k, device=DEV)
    y = torch.randint(0, n, (rows,), device=DEV)
    grads = []
    for hooked in (False, True):
        torch.manual_seed(4)
        lin = torch.nn.Linear(k, n).to(DEV)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        if hooked:
            cl.install()
        try:
            with torch.autocast("cuda", dtype=torch.float16):
                if hooked:
                    assert cl.covers_bf16(x, lin.weight, lin.bias)
                out = lin(x)
                assert out.dtype == torch.float16
                loss = torch.nn.functional.cross_entropy(out.float(), y)
            scaler.scale(loss).backward()
        finally:
            cl.uninstall()
        grads.append((lin.weight.grad.clone() / 1024.0, lin.bias.grad.clone() / 1024.0, out.detach().float()))
    (gw_t, gb_t, o_t), (gw_c, gb_c, o_c) = grads
    assert bool(((o_c - o_t).abs() <= o_t.abs() * 2.0 ** -9 + 1e-3).all())
    assert bool(((gw_c - gw_t).abs() <= 2e-3 * gw_t.abs().max()).all()) and bool(((gb_c - gb_t).abs() <= 2e-3 * gb_t.abs().max()).all())
