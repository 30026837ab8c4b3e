"""The attention projections of GATLayer.forward, (a_l * h).sum(-1) and (a_r * h).sum(-1) (cogdl/layers/gat_layer.py:65-66),
as cogdl_amd.fused._HeadProjections: forward = the layer's own expression; backward against float64 autograd of that
expression -- the fall-back path (few rows / CPU tensors: two-stage column sums) here on the CPU, the MFMA path (parameter
gradients as diagonal blocks of one tall-skinny product, cogdl_hip_linear_wgrad_f32) on the GPU."""
import pytest
import torch

from cogdl_amd.fused import _HeadProjections


def _check(n, h, f, dtype, device, tol):
    gen = torch.Generator().manual_seed(n + h + f)
    feat0 = torch.randn(n, h, f, generator=gen).to(dtype)
    al0, ar0 = torch.randn(1, h, f, generator=gen) * 0.3, torch.randn(1, h, f, generator=gen) * 0.3
    gl, gr = torch.randn(n, h, generator=gen), torch.randn(n, h, generator=gen)
    # float64 autograd of the layer's expression on the same (rounded) operands
    fd, ald, ard = (t.double().requires_grad_() for t in (feat0, al0, ar0))
    ((ald * fd).sum(-1) * gl.double()).sum().backward(retain_graph=True)
    ((ard * fd).sum(-1) * gr.double()).sum().backward()
    feat, al, ar = (t.to(device).requires_grad_() for t in (feat0, al0, ar0))
    hl, hr = _HeadProjections.apply(al, ar, feat)
    # (GPU: one hand-written pass, fp32 products summed left to right; torch's reduction order may differ: fp32 rounding apart)
    for got, a in ((hl, al), (hr, ar)):
        want = (a * feat).sum(-1)
        scale = (a.abs() * feat.abs().float()).sum(-1)
        assert got.dtype == want.dtype and bool(((got - want).abs() <= 1e-6 * scale + 1e-30).all())
    torch.autograd.backward([hl, hr], [gl.to(device), gr.to(device)])
    scale_a = (gl.abs().double().unsqueeze(-1) * feat0.double().abs()).sum(0)
    for got, want, sc in ((al.grad, ald.grad, scale_a), (ar.grad, ard.grad, (gr.abs().double().unsqueeze(-1) * feat0.double().abs()).sum(0))):
        assert got.shape == (1, h, f) and got.dtype == torch.float32
        assert bool(((got.cpu().double() - want).abs() <= 1e-5 * sc + 1e-6).all())
    assert feat.grad.dtype == dtype
    assert bool(((feat.grad.cpu().double() - fd.grad).abs() <= tol * fd.grad.abs() + 1e-6).all())


@pytest.mark.parametrize("n,h,f", [(100, 8, 8), (1500, 1, 41), (7, 3, 5)])
def test_fallback_path_matches_float64_autograd(n, h, f):
    _check(n, h, f, torch.float32, "cpu", 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("n,h,f", [(232965, 8, 8), (50000, 1, 41), (5000, 4, 16), (4097, 3, 5), (300, 8, 8)])
def test_mfma_path_matches_float64_autograd(n, h, f, dtype):
    _check(n, h, f, dtype, "cuda:0", 1e-6 if dtype == torch.float32 else 2.0 ** -8)


def test_an_unused_projection_gets_a_zero_gradient():
    feat = torch.randn(50, 2, 3, requires_grad=True)
    a_l, a_r = torch.randn(1, 2, 3, requires_grad=True), torch.randn(1, 2, 3, requires_grad=True)
    h_l, _ = _HeadProjections.apply(a_l, a_r, feat)
    h_l.sum().backward()
    assert torch.allclose(a_l.grad, feat.detach().sum(0, keepdim=True)) and bool((a_r.grad == 0).all())
    assert torch.allclose(feat.grad, a_l.detach().expand_as(feat))
