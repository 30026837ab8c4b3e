"""The dense side of a CogDL layer: `self.linear(x)` = torch.nn.Linear, whose backward
torch sends to hipBLASLt.  For the tall-skinny shapes of full-graph training (x: [num_nodes, in], grad: [num_nodes, out])
the weight gradient  grad_out^T . x  is a reduction over 10^5..10^8 rows that hipBLASLt runs on a handful of workgroups
(ogbn-arxiv-shaped GCN: 349 + 442 us of a 2.2 ms epoch); `cogdl_hip_linear_wgrad_f32` streams both operands once through
fp32 MFMA instead.

    linear(x, weight, bias)        functional form with the hand-written weight/bias gradient
    install() / uninstall()        make torch.nn.functional.linear -- hence every unchanged nn.Linear inside CogDL's
                                   layers -- take it for the shapes it covers (2-D fp32 GPU input with many rows; under
                                   bf16 autocast: LinearBf16Function, <= 64 output features);
                                   everything else goes to torch's own implementation, as before.
    matmul(x, W)                   `torch.matmul(x, self.W)` of a layer (gat_layer.py:59) under bf16 autocast: the bf16 MFMA
                                   product that reads x once in the dtype the model holds it (csrc/linear_fwd16.hip)
The forward product and grad_input go through `cogdl_hip_linear_fwd_f32` (LDS-staged MFMA, weights resident in LDS)
where that kernel is the faster one (<= 64 output columns, weight <= 96 KB), else stay torch.addmm / torch.mm.
"""
import torch

from . import _lib

MIN_ROWS = 4096      # below this the reduction is too short to matter
MAX_FEATURES = 4096  # in/out features covered by the kernel's tiling

_orig_linear = torch.nn.functional.linear


def linear_wgrad(x, grad_out, want_bias=True):
    """(grad_weight [out, in], grad_bias [out] | None) for y = x @ W^T + b; x [K, in], grad_out [K, out], fp32, GPU."""
    dev = _lib.require_cuda(x, grad_out)
    x, grad_out = x.contiguous(), grad_out.contiguous()
    k, in_f = x.shape
    out_f = grad_out.shape[1]
    grad_w = torch.empty((out_f, in_f), dtype=torch.float32, device=dev)
    grad_b = torch.empty(out_f, dtype=torch.float32, device=dev) if want_bias else None
    ws, ws_bytes = _lib.workspace("cogdl_hip_linear_wgrad_workspace_bytes", dev, k, in_f, out_f)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_linear_wgrad_f32(_lib.ptr(x), _lib.ptr(grad_out), _lib.ptr(grad_w), _lib.ptr(grad_b), k,
                                                   in_f, out_f, _lib.ptr(ws), ws_bytes, _lib.stream_of(x))
    _lib.check(rc, "linear_wgrad")
    return grad_w, grad_b


EUNSUPPORTED = _lib.EUNSUPPORTED  # shape outside the hand-written kernel's coverage: torch's product is used


def tall_skinny_matmul(x, w, bias, w_is_n_by_k):
    """x[R, K] . B (+ bias) through cogdl_hip_linear_fwd_f32, B = w^T (w [N, K]) or w ([K, N]); None if the kernel
    declines the shape (the caller then uses torch's own product)."""
    dev = x.device
    x, w = x.contiguous(), w.contiguous()
    rows, k = x.shape
    n = w.shape[0] if w_is_n_by_k else w.shape[1]
    out = torch.empty((rows, n), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_linear_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), rows, k, n,
                                                 1 if w_is_n_by_k else 0, _lib.stream_of(x))
    if rc == EUNSUPPORTED:
        return None
    _lib.check(rc, "linear_fwd")
    return out


class LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        out = tall_skinny_matmul(x, weight, bias, True)
        return out if out is not None else _orig_linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_x = None
        if ctx.needs_input_grad[0]:
            grad_x = tall_skinny_matmul(grad_out, weight, None, False)
            if grad_x is None:
                grad_x = grad_out.mm(weight)
        grad_w = grad_b = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            grad_w, grad_b = linear_wgrad(x, grad_out, want_bias=need_b)
        elif need_b:
            grad_b = grad_out.sum(0)
        return grad_x, grad_w, grad_b


_HALF_ENTRY = {torch.bfloat16: "cogdl_hip_linear_fwd_bf16", torch.float16: "cogdl_hip_linear_fwd_f16"}


def tall_skinny_matmul_16(x, w, bias, w_is_n_by_k, out_dtype=torch.bfloat16):
    """h(x[R, K]) . h(B) (+ bias) -> [R, N] in `out_dtype` (bfloat16 | float16; h = the rounding to it) through
    cogdl_hip_linear_fwd_bf16 / _f16 (x, w: fp32 or `out_dtype`; B = w^T for w [N, K], w for w [K, N]); None if the kernel
    declines the shape."""
    dev = x.device
    x, w = x.contiguous(), w.contiguous()
    rows, k = x.shape
    n = w.shape[0] if w_is_n_by_k else w.shape[1]
    out = torch.empty((rows, n), dtype=out_dtype, device=dev)
    with _lib.on_device(dev):
        rc = getattr(_lib.hip(), _HALF_ENTRY[out_dtype])(_lib.ptr(x), _lib.DTYPE_CODE[x.dtype], _lib.ptr(w), _lib.DTYPE_CODE[w.dtype],
                                                         _lib.ptr(bias), _lib.ptr(out), rows, k, n, 1 if w_is_n_by_k else 0,
                                                         _lib.stream_of(x))
    if rc == EUNSUPPORTED:
        return None
    _lib.check(rc, "linear_fwd_16")
    return out


def tall_skinny_matmul_bf16(x, w, bias, w_is_n_by_k):
    return tall_skinny_matmul_16(x, w, bias, w_is_n_by_k, torch.bfloat16)


def _dgrad_16(grad_out, w, w_is_n_by_k):
    """grad_input of the 16-bit products through the same kernel where it covers the shape (<= 64 input features, 16-bit rows
    of even length), in grad_out's dtype like torch's product; else None."""
    n = w.shape[0] if w_is_n_by_k else w.shape[1]
    if grad_out.dtype not in _HALF_ENTRY or n > 64 or grad_out.shape[0] < MIN_ROWS or w.dtype not in (torch.float32, grad_out.dtype):
        return None
    return tall_skinny_matmul_16(grad_out, w, None, w_is_n_by_k, grad_out.dtype)


def _autocast_half():
    """The 16-bit dtype of the enclosing torch.autocast("cuda", ...) context (bfloat16 | float16), or None."""
    if not torch.is_autocast_enabled("cuda"):
        return None
    dt = torch.get_autocast_dtype("cuda")
    return dt if dt in _HALF_ENTRY else None


class MatmulBf16Function(torch.autograd.Function):
    """torch.matmul(x, W) of a layer under bf16 autocast (x [R, K] tall, W [K, N] with N <= 64), round 6.
    forward: cogdl_hip_linear_fwd_bf16 -- x is read ONCE, in the dtype the model holds it (fp32 x is rounded to bf16 in
        registers: autocast's cast kernel and its bf16 copy of x do not exist), bf16 result like autocast's;
    grad_W:  the fp32 split-K MFMA reduction of linear_wgrad over the SAVED INPUT ITSELF (autocast saves the bf16 copy and
        sends hipBLASLt a [N, R] x [R, K] bf16 product it runs at 0.5 TB/s): x^T . fp32(grad_out), fp32 accumulation of
        un-rounded x -- at least as accurate as autocast's product;
    grad_x:  grad_out . W^T in grad_out's dtype (torch; only layers behind the first need it, where x is narrow)."""

    @staticmethod
    def forward(ctx, x, w, half=torch.bfloat16):
        out = tall_skinny_matmul_16(x, w, None, False, half)
        if out is None:  # (a shape the kernel declines: the same product by torch, same backward)
            out = torch.mm(x.to(half), w.to(half))
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, w = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_x = grad_w = None
        if ctx.needs_input_grad[0]:
            grad_x = _dgrad_16(grad_out, w, True)  # grad_out . W^T, W stored [K, N] = [n, k] of the product
            if grad_x is None:
                grad_x = torch.mm(grad_out, w.to(grad_out.dtype).t())
            grad_x = grad_x.to(x.dtype)
        if ctx.needs_input_grad[1]:
            gw, _ = linear_wgrad(x.float(), grad_out.float(), want_bias=False)  # [N, K] = grad_out^T . x
            grad_w = gw.t().to(w.dtype)
        return grad_x, grad_w, None


def matmul_covers(x, w):
    return (torch.is_tensor(x) and torch.is_tensor(w) and x.is_cuda and w.is_cuda and w.device == x.device
            and x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[0] and x.shape[0] >= MIN_ROWS
            and w.shape[1] <= 64 and w.shape[0] <= MAX_FEATURES
            and _autocast_half() is not None
            and x.dtype in (torch.float32, _autocast_half()) and w.dtype in (torch.float32, _autocast_half()))


def matmul(x, w):
    """Drop-in for `torch.matmul(x, self.W)` in a layer's forward (cogdl/layers/gat_layer.py:59): under bf16 autocast and for
    the tall-skinny shapes of full-graph training MatmulBf16Function, else torch's own product."""
    if matmul_covers(x, w):
        return MatmulBf16Function.apply(x, w, _autocast_half())
    return torch.matmul(x, w)


class LinearBf16Function(torch.autograd.Function):
    """torch.nn.functional.linear(x, W, b) under bf16 autocast for tall x and <= 64 output features: forward by
    cogdl_hip_linear_fwd_bf16 (B = W^T; the bias rounded to bf16 like autocast's cast, added in fp32 before the result is
    rounded), grad_W / grad_b by the fp32 split-K MFMA reduction over the saved input, grad_x by torch."""

    @staticmethod
    def forward(ctx, x, weight, bias, half=torch.bfloat16):
        b = None if bias is None else bias.detach().to(half).float()
        out = tall_skinny_matmul_16(x, weight, b, True, half)
        if out is None:
            out = _orig_linear(x.to(half), weight.to(half), None if bias is None else bias.to(half))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_x = grad_w = grad_b = None
        if ctx.needs_input_grad[0]:
            grad_x = _dgrad_16(grad_out, weight, False)  # grad_out . W, W stored [N, K] = [k, n] of the product
            if grad_x is None:
                grad_x = torch.mm(grad_out, weight.to(grad_out.dtype))
            grad_x = grad_x.to(x.dtype)
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            grad_w, grad_b = linear_wgrad(x.float(), grad_out.float(), want_bias=need_b)
            grad_w = grad_w.to(weight.dtype)
        elif need_b:
            grad_b = grad_out.float().sum(0)
        if grad_b is not None:
            grad_b = grad_b.to(ctx.bias_dtype)
        return grad_x, grad_w, grad_b, None


def covers_bf16(x, weight, bias):
    """nn.Linear under torch.autocast("cuda", bfloat16 | float16 -- the latter is the reference's Trainer(fp16=True)) on the
    tall-skinny shapes of full-graph training."""
    return (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and weight.dim() == 2 and weight.is_cuda and weight.device == x.device
            and x.shape[1] == weight.shape[1] and x.shape[0] >= MIN_ROWS and weight.shape[0] <= 64 and weight.shape[1] <= MAX_FEATURES
            and _autocast_half() is not None
            and x.dtype in (torch.float32, _autocast_half()) and weight.dtype in (torch.float32, _autocast_half())
            and (bias is None or (bias.is_cuda and bias.device == x.device and bias.dim() == 1 and bias.shape[0] == weight.shape[0]
                                  and bias.dtype in (torch.float32, _autocast_half()))))


def covers(x, weight, bias):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and x.shape[0] >= MIN_ROWS and weight.dim() == 2 and max(weight.shape) <= MAX_FEATURES
            # anything torch would reject (shape or device mismatch) must reach torch so that IT raises
            and x.shape[1] == weight.shape[1] and weight.is_cuda and weight.device == x.device
            and (bias is None or (bias.dtype == torch.float32 and bias.is_cuda and bias.device == x.device
                                  and bias.dim() == 1 and bias.shape[0] == weight.shape[0]))
            and torch.is_grad_enabled() and weight.requires_grad and not torch.is_autocast_enabled("cuda"))


def linear(x, weight, bias=None):
    """Drop-in for torch.nn.functional.linear."""
    if covers(x, weight, bias):
        return LinearFunction.apply(x, weight, bias)
    if covers_bf16(x, weight, bias):
        return LinearBf16Function.apply(x, weight, bias, _autocast_half())
    return _orig_linear(x, weight, bias)


def install():
    """Idempotent: torch.nn.functional.linear (what nn.Linear.forward calls) -> `linear` above."""
    _lib.hip()  # fail loudly if the library is missing
    torch.nn.functional.linear = linear


def uninstall():
    torch.nn.functional.linear = _orig_linear
