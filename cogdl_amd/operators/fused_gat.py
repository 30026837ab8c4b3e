"""Drop-in for cogdl/operators/fused_gat.py: `fused_gat_func(attn_row, attn_col, row_ptr, col_ind, col_ptr,
row_ind, negative_slope, in_feat)` -> [N, H, F] (operators/fused_gat.py:6-41).

The reference forwards to dgNN's `fused_gatconv` (an empty submodule in the tree); CogDL only takes that path
when `attn_drop == 0 and graph.is_symmetric()` (layers/gat_layer.py:68) and then passes the CSR twice, as
(row_ptr, col_ind) and again as the "CSC" (utils/spmm_utils.py:258-261).  Here col_ptr/row_ind are accepted
for signature compatibility but NOT trusted: the backward uses the true, cached transpose of (row_ptr,
col_ind), so the op is also correct for non-symmetric graphs and sampled blocks.
Semantics = the unfused path: edge_softmax(LeakyReLU(attn_row[row] + attn_col[col])) then mh_spmm.
"""
import torch

from .. import _lib
from ..plan import PLANS, Fingerprint

_lib.hip()


def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    dev = _lib.require_cuda(attn_row, attn_col, row_ptr, col_ind, in_feat)
    if in_feat.dim() != 3 or in_feat.dtype not in _lib.DTYPE_CODE:
        raise _lib.BackendError("in_feat must be [N, H, F] float32/float16/bfloat16")
    feat = in_feat.contiguous()
    attn_row, attn_col = attn_row.contiguous().float(), attn_col.contiguous().float()
    v, (n_src, h, f) = row_ptr.numel() - 1, feat.shape
    if attn_row.shape != (v, h) or attn_col.shape != (n_src, h):
        raise _lib.BackendError("attn_row/attn_col must be [N_dst, H]/[N_src, H]")
    out = torch.empty((v, h, f), dtype=feat.dtype, device=dev)
    edge_max = torch.empty((v, h), dtype=torch.float32, device=dev)
    edge_sum = torch.empty((v, h), dtype=torch.float32, device=dev)
    nnz, code = col_ind.numel(), _lib.DTYPE_CODE[feat.dtype]
    ws, ws_bytes = _lib.workspace("cogdl_hip_gat_fwd_workspace_bytes", dev, nnz, h, f, code)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_gat_fwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(attn_row),
                                          _lib.ptr(attn_col), _lib.ptr(feat), float(negative_slope), _lib.ptr(out),
                                          _lib.ptr(edge_max), _lib.ptr(edge_sum), v, h, f, nnz, code,
                                          _lib.ptr(ws), ws_bytes, _lib.stream_of(feat))
    _lib.check(rc, "gat_fwd")
    return out, edge_max, edge_sum


def _unfused_backward(negative_slope, row_ptr, col_ind, in_feat, attn_row, attn_col, grad_out):
    """Shapes the fused backward does not cover: differentiate the composition of the unfused HIP operators."""
    from .edge_softmax import csr_edge_softmax
    from .mhspmm import csrmhspmm

    with torch.enable_grad():
        ar, ac, ft = (t.detach().float().requires_grad_() for t in (attn_row, attn_col, in_feat))
        deg = (row_ptr[1:] - row_ptr[:-1]).long()
        row = torch.repeat_interleave(torch.arange(deg.numel(), device=deg.device), deg)
        score = torch.nn.functional.leaky_relu(ar[row] + ac[col_ind.long()], negative_slope)
        out = csrmhspmm(row_ptr, col_ind, ft, csr_edge_softmax(row_ptr, score))
        g_ar, g_ac, g_ft = torch.autograd.grad(out, (ar, ac, ft), grad_out.float())
    return g_ft, g_ar, g_ac


class FusedGATFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat):
        row_ptr, col_ind = _lib.csr_structure(row_ptr, col_ind)
        ctx.fp = Fingerprint(row_ptr, col_ind, in_feat.shape[0])  # before the kernel: lands early for backward
        out, edge_max, edge_sum = gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)
        ctx.save_for_backward(row_ptr, col_ind, edge_max, edge_sum, in_feat, attn_row, attn_col, out)
        ctx.negative_slope = float(negative_slope)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        row_ptr, col_ind, edge_max, edge_sum, in_feat, attn_row, attn_col, out = ctx.saved_tensors
        dev = grad_out.device
        v, (n_src, h, f) = row_ptr.numel() - 1, in_feat.shape
        # feat / out / grad_out are read in the layer's dtype (bf16 for configs[2]): no fp32 copies
        dt = in_feat.dtype
        g = grad_out.contiguous() if grad_out.dtype == dt else grad_out.to(dt).contiguous()
        feat, o = in_feat.detach().contiguous(), out.detach().contiguous()
        ar, ac = attn_row.detach().contiguous().float(), attn_col.detach().contiguous().float()
        plan = PLANS.get(ctx.fp, row_ptr, col_ind, n_src)
        grad_feat = torch.empty((n_src, h, f), dtype=dt, device=dev)
        grad_ar = torch.empty((v, h), dtype=torch.float32, device=dev)
        grad_ac = torch.empty((n_src, h), dtype=torch.float32, device=dev)
        lib = _lib.hip()
        nnz, code = col_ind.numel(), _lib.DTYPE_CODE[dt]
        ws_bytes = lib.cogdl_hip_gat_bwd_workspace_bytes(v, h, f, nnz, code)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_gat_bwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(plan.colptr),
                                       _lib.ptr(plan.rowind), _lib.ptr(ar), _lib.ptr(ac), _lib.ptr(feat),
                                       ctx.negative_slope, _lib.ptr(edge_max), _lib.ptr(edge_sum), _lib.ptr(o),
                                       _lib.ptr(g), _lib.ptr(grad_feat), _lib.ptr(grad_ar), _lib.ptr(grad_ac),
                                       _lib.ptr(ws), ws_bytes, v, n_src, h, f, nnz, code, _lib.stream_of(g))
        if rc == _lib.EUNSUPPORTED:  # a valid call whose shape the fused backward declines (any other status raises)
            grad_feat, grad_ar, grad_ac = _unfused_backward(ctx.negative_slope, row_ptr, col_ind, in_feat, attn_row,
                                                            attn_col, grad_out)
        else:
            _lib.check(rc, "gat_bwd")
        return (grad_ar.to(attn_row.dtype), grad_ac.to(attn_col.dtype), None, None, None, None, None,
                grad_feat.to(in_feat.dtype))


def fused_gat_func(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat):
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat)
