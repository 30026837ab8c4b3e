"""Drop-in for cogdl/operators/edge_softmax.py: `csr_edge_softmax(rowptr, h)` -> [E, H]
(operators/edge_softmax.py:17-38), any number of heads."""
import torch

from .. import _lib

_lib.hip()


def _launch(fn_name, rowptr, a, g=None):
    dev = _lib.require_cuda(rowptr, a, g)
    if rowptr.dtype != torch.int32:
        raise _lib.BackendError("rowptr must be int32")
    if a.dim() != 2 or a.dtype != torch.float32:
        raise _lib.BackendError("edge values must be a float32 [E, H] tensor, got %s %s" % (a.dtype, tuple(a.shape)))
    a = a.contiguous()
    out = torch.empty_like(a)
    m, (nnz, h) = rowptr.numel() - 1, a.shape
    fn = getattr(_lib.hip(), fn_name)
    ws, ws_bytes = _lib.workspace("cogdl_hip_edge_softmax_workspace_bytes", dev, nnz, h)
    with _lib.on_device(dev):
        if g is None:
            rc = fn(_lib.ptr(rowptr), _lib.ptr(a), _lib.ptr(out), m, nnz, h, _lib.ptr(ws), ws_bytes,
                    _lib.stream_of(a))
        else:
            g = g.contiguous()
            rc = fn(_lib.ptr(rowptr), _lib.ptr(a), _lib.ptr(g), _lib.ptr(out), m, nnz, h, _lib.ptr(ws), ws_bytes,
                    _lib.stream_of(a))
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
        return None, _launch("cogdl_hip_edge_softmax_bwd", rowptr, out, grad_out.contiguous().float())


def csr_edge_softmax(rowptr, h):
    return EdgeSoftmaxFunction.apply(rowptr, h)
