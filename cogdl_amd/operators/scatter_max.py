"""Drop-in for cogdl/operators/scatter_max.py: `scatter_max(rowptr, colind, feat)` -> [M, F]
(operators/scatter_max.py:17-37), used by MaxAggregator (cogdl/layers/sage_layer.py:21-29).

Two result conventions:
  * default: the true segment maximum (rows whose values are all negative get their real maximum; an empty row gives 0);
  * reference-exact (`scatter_max(..., reference_exact=True)` or COGDL_AMD_SCATTER_MAX_REFERENCE=1): what the reference's
    CUDA kernel returns -- its accumulator starts at FLT_MIN, the smallest positive normal float
    (operators/scatter_max/scatter_max.cu:16), so every non-empty row's result is max(FLT_MIN, true maximum), and
    where nothing exceeds FLT_MIN the argmax is undefined there (here: -1, no gradient flows).  The two agree whenever
    a row's maximum exceeds FLT_MIN.
"""
import os

import torch

from .. import _lib
from ..plan import PLANS, Fingerprint, fingerprint_of

_lib.hip()


def scatter_max_fp(rowptr, colind, feat):
    dev = _lib.require_cuda(rowptr, colind, feat)
    if rowptr.dtype != torch.int32 or colind.dtype != torch.int32:
        raise _lib.BackendError("rowptr/colind must be int32")
    if feat.dim() != 2 or feat.dtype != torch.float32:
        raise _lib.BackendError("feat must be a float32 [N, F] tensor")
    feat = feat.contiguous()
    m, k = rowptr.numel() - 1, feat.shape[1]
    out = torch.empty((m, k), dtype=torch.float32, device=dev)
    max_id = torch.empty((m, k), dtype=torch.int32, device=dev)
    nnz = colind.numel()
    ws, ws_bytes = _lib.workspace("cogdl_hip_scatter_max_workspace_bytes", dev, nnz, k)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_scatter_max_fwd(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(feat), _lib.ptr(out),
                                                  _lib.ptr(max_id), m, k, nnz, _lib.ptr(ws), ws_bytes,
                                                  _lib.stream_of(feat))
    _lib.check(rc, "scatter_max_fwd")
    return out, max_id


def scatter_max_bp(grad, max_id, n_src):
    """The reference's formulation (one fp32 atomic per element into a zeroed buffer); kept for callers without the
    graph structure at hand.  The autograd Function uses scatter_max_bp_csc."""
    dev = _lib.require_cuda(grad, max_id)
    grad = grad.contiguous().float()
    m, k = grad.shape
    out = torch.empty((n_src, k), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_scatter_max_bwd(_lib.ptr(grad), _lib.ptr(max_id), _lib.ptr(out), m, k, n_src,
                                                  _lib.stream_of(grad))
    _lib.check(rc, "scatter_max_bwd")
    return out


def scatter_max_bp_csc(colptr, rowind, grad, max_id, n_src):
    """The backward as a gather over the transposed structure: no atomics, no zero-fill, deterministic (and equal to
    the sequential reference loop bit for bit while a source node has at most `long-row threshold` out-edges)."""
    dev = _lib.require_cuda(colptr, rowind, grad, max_id)
    grad = grad.contiguous().float()
    k, nnz = grad.shape[1], rowind.numel()
    out = torch.empty((n_src, k), dtype=torch.float32, device=dev)
    ws, ws_bytes = _lib.workspace("cogdl_hip_scatter_max_bwd_workspace_bytes", dev, nnz, k)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_scatter_max_bwd_csc(_lib.ptr(colptr), _lib.ptr(rowind), _lib.ptr(grad),
                                                      _lib.ptr(max_id), _lib.ptr(out), n_src, k, nnz, _lib.ptr(ws),
                                                      ws_bytes, _lib.stream_of(grad))
    _lib.check(rc, "scatter_max_bwd_csc")
    return out


FLT_MIN = 1.1754943508222875e-38


class ScatterMaxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowptr, colind, feat, reference_exact=False):
        rowptr, colind = _lib.csr_structure(rowptr, colind)
        ctx.fp = fingerprint_of(rowptr, colind, feat.shape[0]) if ctx.needs_input_grad[2] else None  # before the kernel
        out, max_id = scatter_max_fp(rowptr, colind, feat)
        if reference_exact:  # acc = FLT_MIN; if (acc < B) {acc = B; max_id = cid}  (scatter_max.cu:16-23)
            below = ((rowptr[1:] > rowptr[:-1]).view(-1, 1)) & ~(out > FLT_MIN)
            out = torch.where(below, torch.full_like(out, FLT_MIN), out)
            max_id = torch.where(below, torch.full_like(max_id, -1), max_id)
        ctx.save_for_backward(max_id, rowptr, colind)
        ctx.n_src = feat.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad):
        max_id, rowptr, colind = ctx.saved_tensors
        # the cached transpose of the structure (the plan SpMM's backward uses, too) turns the scatter into a gather
        plan = PLANS.get(ctx.fp, rowptr, colind, ctx.n_src)
        return None, None, scatter_max_bp_csc(plan.colptr, plan.rowind, grad, max_id, ctx.n_src), None


def scatter_max(rowptr, colind, feat, reference_exact=None):
    if reference_exact is None:
        reference_exact = os.environ.get("COGDL_AMD_SCATTER_MAX_REFERENCE", "0") == "1"
    return ScatterMaxFunction.apply(rowptr, colind, feat, bool(reference_exact))
