"""Drop-in for cogdl/operators/edge_softmax.py: `csr_edge_softmax(rowptr, h)` -> [E, H]
(operators/edge_softmax.py:17-38), any number of heads; float32, float16 or bfloat16 edge values (fp32 arithmetic,
the result has the dtype of the input -- what GATLayer needs for configs[2]'s bf16 training)."""
import torch

from .. import _lib

_lib.hip()

# The workspace of the most recent call (tests read its header: word 0 counts the cross-tile waits of the flat kernel
# that timed out and recomputed their row statistics -- 0 in normal operation).
LAST_WORKSPACE = None


def _aligned(t):
    """Contiguous and 16-byte aligned (a fresh allocation always is; a view at an odd offset is copied)."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def _launch(fn_name, rowptr, a, g=None):
    dev = _lib.require_cuda(rowptr, a, g)
    if rowptr.dtype != torch.int32:
        raise _lib.BackendError("rowptr must be int32")
    if a.dim() != 2 or a.dtype not in _lib.DTYPE_CODE:
        raise _lib.BackendError("edge values must be a float32/float16/bfloat16 [E, H] tensor, got %s %s"
                                % (a.dtype, tuple(a.shape)))
    rowptr = rowptr.contiguous()
    a = _aligned(a)
    m, (nnz, h) = rowptr.numel() - 1, a.shape
    lib = _lib.hip()
    fn = getattr(lib, fn_name)
    global LAST_WORKSPACE
    ws, ws_bytes = _lib.workspace("cogdl_hip_edge_softmax_workspace_bytes", dev, nnz, h)
    LAST_WORKSPACE = ws
    if g is not None:
        g = _aligned(g if g.dtype == a.dtype else g.to(a.dtype))

    def call(a_, g_, out_, code):
        with _lib.on_device(dev):
            if g_ is None:
                return fn(_lib.ptr(rowptr), _lib.ptr(a_), _lib.ptr(out_), m, nnz, h, code, _lib.ptr(ws), ws_bytes,
                          _lib.stream_of(a_))
            return fn(_lib.ptr(rowptr), _lib.ptr(a_), _lib.ptr(g_), _lib.ptr(out_), m, nnz, h, code, _lib.ptr(ws),
                      ws_bytes, _lib.stream_of(a_))

    out = torch.empty_like(a)
    rc = call(a, g, out, _lib.DTYPE_CODE[a.dtype])  # (every dtype x every H is a kernel of its own: no conversion route)
    _lib.check(rc, fn_name)
    return out


class EdgeSoftmaxFunction(torch.autograd.Function):
    """Mirrors cogdl.operators.edge_softmax.EdgeSoftmaxFunction (operators/edge_softmax.py:26-38)."""

    @staticmethod
    def forward(ctx, rowptr, h):
        out = _launch("cogdl_hip_edge_softmax_fwd", rowptr, h)
        ctx.save_for_backward(rowptr, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, out = ctx.saved_tensors
        return None, _launch("cogdl_hip_edge_softmax_bwd", rowptr, out, grad_out)


def csr_edge_softmax(rowptr, h):
    return EdgeSoftmaxFunction.apply(rowptr, h)
