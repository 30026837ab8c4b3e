"""Drop-in for cogdl/operators/mhspmm.py: `csrmhspmm(rowptr, colind, feat[N,H,F], attention[E,H])`
-> [N,H,F] (operators/mhspmm.py:34-64).

Backward (same maths as MHSPMMFunction.backward, operators/mhspmm.py:52-64):
    grad_feat = mhspmm(A^T, attention[perm], grad_out)      -- A^T/perm from the cached plan, the permutation
                applied inside the kernel (cogdl_hip_mhspmm_eid); the
                reference transposes arange(nnz) through a float32 cuSPARSE call, which loses
                edge ids above 2^24 (Reddit has 1.1e8 edges); here perm is int32 throughout.
    grad_att  = mhsddmm(A, grad_out, feat)
"""
import torch

from .. import _lib
from ..plan import PLANS, Fingerprint, fingerprint_of

_lib.hip()


def mhspmm_raw(rowptr, colind, att, feat, eid=None):
    """out[v,h,:] = sum_e att[eid[e] | e, h] * feat[colind[e],h,:]"""
    dev = _lib.require_cuda(rowptr, colind, att, feat, eid)
    if feat.dim() != 3:
        raise _lib.BackendError("feat must be [N, H, F], got %s" % (tuple(feat.shape),))
    if feat.dtype not in _lib.DTYPE_CODE:
        raise _lib.BackendError("unsupported dtype %s" % feat.dtype)
    feat = feat.contiguous()
    att = att.contiguous().float()
    v, (_, h, f) = rowptr.numel() - 1, feat.shape
    if att.shape != (colind.numel(), h):
        raise _lib.BackendError("attention must be [E, H] = %s, got %s" % ((colind.numel(), h), tuple(att.shape)))
    out = torch.empty((v, h, f), dtype=feat.dtype, device=dev)
    ws, ws_bytes = _lib.workspace("cogdl_hip_mhspmm_workspace_bytes", dev, colind.numel(), h, f,
                                  _lib.DTYPE_CODE[feat.dtype])
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_mhspmm_eid(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(att), _lib.ptr(eid),
                                             _lib.ptr(feat), _lib.ptr(out), v, h, f, colind.numel(),
                                             _lib.DTYPE_CODE[feat.dtype], _lib.ptr(ws), ws_bytes, _lib.stream_of(feat))
    _lib.check(rc, "mhspmm")
    return out


def mhsddmm_raw(rowptr, colind, grad, feat):
    dev = _lib.require_cuda(rowptr, colind, grad, feat)
    grad, feat = grad.contiguous().float(), feat.contiguous().float()
    v, (_, h, f) = rowptr.numel() - 1, feat.shape
    nnz = colind.numel()
    out = torch.empty((nnz, h), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_mhsddmm(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(grad), _lib.ptr(feat),
                                          _lib.ptr(out), v, h, f, nnz, _lib.stream_of(feat))
    _lib.check(rc, "mhsddmm")
    return out


class MHSPMMFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowptr, colind, feat, attention):
        rowptr, colind = _lib.csr_structure(rowptr, colind)
        ctx.fp = fingerprint_of(rowptr, colind, feat.shape[0]) if ctx.needs_input_grad[2] else None  # before the kernel
        out = mhspmm_raw(rowptr, colind, attention, feat)
        ctx.save_for_backward(rowptr, colind, feat, attention)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, colind, feat, attention = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_feat = grad_att = None
        if ctx.needs_input_grad[2]:
            plan = PLANS.get(ctx.fp, rowptr, colind, feat.shape[0])
            # A^T with the attention left in CSR order: the kernel reads att[perm[j]] (no transposed [E, H] copy)
            grad_feat = mhspmm_raw(plan.colptr, plan.rowind, attention.detach(), grad_out.to(feat.dtype), eid=plan.perm)
        if ctx.needs_input_grad[3]:
            grad_att = mhsddmm_raw(rowptr, colind, grad_out, feat.detach()).to(attention.dtype)
        return None, None, grad_feat, grad_att


def csrmhspmm(rowptr, colind, feat, attention):
    return MHSPMMFunction.apply(rowptr, colind, feat, attention)
