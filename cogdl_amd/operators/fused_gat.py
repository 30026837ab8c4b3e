"""Drop-in for cogdl/operators/fused_gat.py: `fused_gat_func(attn_row, attn_col, row_ptr, col_ind, col_ptr,
row_ind, negative_slope, in_feat)` -> [N, H, F] (operators/fused_gat.py:6-41).

The reference forwards to dgNN's `fused_gatconv` (an empty submodule in the tree); CogDL only takes that path
when `attn_drop == 0 and graph.is_symmetric()` (layers/gat_layer.py:68) and then passes the CSR twice, as
(row_ptr, col_ind) and again as the "CSC" (utils/spmm_utils.py:258-261).  Here col_ptr/row_ind are accepted
for signature compatibility but NOT trusted: the backward uses the true, cached transpose of (row_ptr,
col_ind), so the op is also correct for non-symmetric graphs and sampled blocks.
Semantics = the unfused path: edge_softmax(LeakyReLU(attn_row[row] + attn_col[col])) then mh_spmm.  Any H x F (the
reference's backward has no shape limit, operators/fused_gat.py:28-40).

`fused_gat_dropout_func` adds the attention dropout of the branch CogDL's gat model takes by default (attn_drop 0.5,
models/nn/gat.py:30; layers/gat_layer.py:72-77) inside the same kernels.
"""
import torch

from .. import _lib, xcdplan
from .. import plan as _plan
from ..plan import PLANS, Fingerprint, fingerprint_of

_lib.hip()


def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, p=0.0, seed=0, xplan=None):
    """out [N_dst, H, F], edge_max, edge_sum [N_dst, H].  p > 0: attention dropout inside the kernel, the mask a pure
    function of (seed, edge position, head) -- see include/cogdl_hip.h.  xplan: an XCD-partitioned plan of the structure
    (cogdl_amd/xcdplan.py); shapes its kernels decline fall back to the ordinary entry."""
    return _gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, p, seed, xplan)[:3]


def _gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, p, seed, xplan):
    """gat_forward + whether the plan's kernels ran."""
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
    lib = _lib.hip()
    if xplan is not None:
        ws, ws_bytes = _lib.workspace("cogdl_hip_gat_fwd_xcd_workspace_bytes", dev, xplan.n_parts, h, f, code)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_gat_fwd_xcd(xplan.ref(), _lib.ptr(attn_row), _lib.ptr(attn_col), _lib.ptr(feat),
                                           float(negative_slope), float(p), int(seed), _lib.ptr(out), _lib.ptr(edge_max),
                                           _lib.ptr(edge_sum), v, h, f, code, _lib.ptr(ws), ws_bytes, _lib.stream_of(feat))
        if rc != _lib.EUNSUPPORTED:
            _lib.check(rc, "gat_fwd_xcd")
            return out, edge_max, edge_sum, True
    ws, ws_bytes = _lib.workspace("cogdl_hip_gat_fwd_workspace_bytes", dev, nnz, h, f, code)
    with _lib.on_device(dev):
        if p > 0.0:
            rc = lib.cogdl_hip_gat_dropout_fwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(attn_row),
                                               _lib.ptr(attn_col), _lib.ptr(feat), float(negative_slope), float(p),
                                               int(seed), _lib.ptr(out), _lib.ptr(edge_max), _lib.ptr(edge_sum), v, h,
                                               f, nnz, code, _lib.ptr(ws), ws_bytes, _lib.stream_of(feat))
        else:
            rc = lib.cogdl_hip_gat_fwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(attn_row), _lib.ptr(attn_col),
                                       _lib.ptr(feat), float(negative_slope), _lib.ptr(out), _lib.ptr(edge_max),
                                       _lib.ptr(edge_sum), v, h, f, nnz, code, _lib.ptr(ws), ws_bytes,
                                       _lib.stream_of(feat))
    _lib.check(rc, "gat_fwd")
    return out, edge_max, edge_sum, False


def edge_dropout_mask(nnz, heads, p, seed, device):
    """d[e,h] of the fused dropout as a dense [nnz, heads] fp32 tensor (0 or the keep scale): what
    `FusedGATFunction` with (p, seed) applies to the attention, for tests and for callers that want the same mask on the
    unfused operators."""
    mask = torch.empty((nnz, heads), dtype=torch.float32, device=device)
    with _lib.on_device(mask.device):
        rc = _lib.hip().cogdl_hip_edge_dropout_mask(nnz, heads, float(p), int(seed), _lib.ptr(mask),
                                                    _lib.stream_of(mask))
    _lib.check(rc, "edge_dropout_mask")
    return mask


PAD_FEATURES = True  # tests switch it off to drive the kernels at the caller's own (odd, unaligned) widths


def _padded_width(f, elem_bytes, heads=1):
    """Feature width the kernels run at.  A row of F elements whose byte length is not a multiple of 16 forces narrow,
    unaligned loads on EVERY gathered row (F = 41 in bf16: 82-byte rows, one 2-byte load per lane: the second layer of
    the gat model ran at 32 % of the roofline); padding feat once per call costs a pass over [N, H, F] -- nothing next to the
    per-edge gathers -- and the padded columns are exact zeros throughout.
    Round 6: rows (H x F elements) between 64 and 256 bytes are padded on to 128 / 256 bytes -- ONE (two) L2 lines per gathered
    row instead of a row that straddles line borders: Reddit-shaped graph, H = 1 bf16, F = 48 (96-byte rows) forward 1920 us /
    backward 4193 us, F = 64 (128-byte rows) 1587 / 3397 us (profiles/r06_gat_h1.txt) -- the gathers are L2 line requests, and
    the lanes the wider row occupies were idle anyway (lane groups are powers of two)."""
    if not PAD_FEATURES:
        return f
    q = 16 // elem_bytes
    fp = (f + q - 1) // q * q
    row = heads * fp * elem_bytes
    if 64 < row <= 256 and row & (row - 1):
        target = 128 if row < 128 else 256
        if target % (heads * elem_bytes) == 0 and (target // (heads * elem_bytes)) % q == 0:
            fp = target // (heads * elem_bytes)
    return fp


class FusedGATFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat, p=0.0, seed=0):
        row_ptr, col_ind = _lib.csr_structure(row_ptr, col_ind)
        ctx.fp = fingerprint_of(row_ptr, col_ind, in_feat.shape[0])  # before the kernel: lands early for backward
        f = in_feat.shape[-1]
        fp = (_padded_width(f, in_feat.element_size(), in_feat.shape[1])
              if in_feat.dim() == 3 and in_feat.dtype in _lib.DTYPE_CODE else f)
        feat = in_feat.detach()
        if fp != f:
            feat = torch.nn.functional.pad(feat, (0, fp - f))
        # XCD-partitioned plan (cogdl_amd/xcdplan.py): hub-heavy graphs over cache-sized tables (BASELINE configs[2])
        ok = feat.dim() == 3 and feat.dtype in _lib.DTYPE_CODE
        row_bytes = feat.shape[1] * fp * feat.element_size() if ok else 0

        def decide():
            xcd = ok and xcdplan.wanted(row_ptr.numel() - 1, col_ind.numel(), feat.shape[0], row_bytes)
            if ok and not xcd:
                # a memoised fingerprint (install(structure_memo=True), or the identity memo of plan.fingerprint_of once a backward
                # pass has asked for its key), or the recorded eager run of cogdl_amd.graphs.capture (which may wait for the
                # hash): skewed structures of any size (xcdplan.ordered_wanted)
                if (getattr(row_ptr, "_cogdl_amd_struct", None) is not None or _plan.recording()) and ctx.fp.event is not None:
                    ctx.fp.key()
                if ctx.fp._key is not None:
                    xcd = xcdplan.ordered_wanted(ctx.fp, row_ptr, row_ptr.numel() - 1, col_ind.numel(), feat.shape[0], row_bytes)
            return xcdplan.csr_plan(ctx.fp, row_ptr, col_ind) if xcd else None

        xplan = _plan.taped_choice("fused_gat.forward", decide)  # (a capture replays the recorded run's plans: plan.PlanTape)
        ctx.xcd = xplan is not None
        # (a shape the plan's forward declines -- column tiles -- keeps the ordinary backward too)
        if xplan is not None:
            out, edge_max, edge_sum, ctx.xcd = _gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, feat, p,
                                                            seed, xplan)
        else:
            out, edge_max, edge_sum = gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, feat, p, seed)
        ctx.save_for_backward(row_ptr, col_ind, edge_max, edge_sum, feat, attn_row, attn_col, out)
        ctx.negative_slope, ctx.p, ctx.seed, ctx.f = float(negative_slope), float(p), int(seed), f
        ctx.feat_dtype = in_feat.dtype
        return out[..., :f].contiguous() if fp != f else out

    @staticmethod
    def backward(ctx, grad_out):
        row_ptr, col_ind, edge_max, edge_sum, feat, attn_row, attn_col, out = ctx.saved_tensors
        dev = grad_out.device
        v, (n_src, h, fp) = row_ptr.numel() - 1, feat.shape
        # feat / out / grad_out are read in the layer's dtype (bf16 for configs[2]): no fp32 copies
        dt = feat.dtype
        g = grad_out if grad_out.dtype == dt else grad_out.to(dt)
        g = torch.nn.functional.pad(g, (0, fp - ctx.f)) if fp != ctx.f else g.contiguous()
        feat, o = feat.contiguous(), out.contiguous()
        ar, ac = attn_row.detach().contiguous().float(), attn_col.detach().contiguous().float()
        plan = PLANS.get(ctx.fp, row_ptr, col_ind, n_src)
        grad_feat = torch.empty((n_src, h, fp), dtype=dt, device=dev)
        grad_ar = torch.empty((v, h), dtype=torch.float32, device=dev)
        grad_ac = torch.empty((n_src, h), dtype=torch.float32, device=dev)
        lib = _lib.hip()
        nnz, code = col_ind.numel(), _lib.DTYPE_CODE[dt]
        rc = _lib.EUNSUPPORTED
        # (the structure's key is known by now -- PLANS.get has waited for the hash: a skewed structure takes the plans in
        #  backward whether or not its forward call could)
        #  -- from the structure's second sighting on: a one-off structure must not pay the plan builds)

        def decide():
            if ctx.xcd or (getattr(plan, "sightings", 1) > 1 and
                           xcdplan.ordered_wanted(ctx.fp, row_ptr, v, col_ind.numel(), n_src, h * fp * feat.element_size())):
                return xcdplan.csr_plan(ctx.fp, row_ptr, col_ind), xcdplan.csc_plan(ctx.fp, plan)
            return None

        xplans = _plan.taped_choice("fused_gat.backward", decide)
        if xplans is not None:
            xr, xc = xplans
            ws, ws_bytes = _lib.workspace("cogdl_hip_gat_bwd_xcd_workspace_bytes", dev, v, h, fp, xr.n_parts, xc.n_parts, code)
            with _lib.on_device(dev):
                rc = lib.cogdl_hip_gat_bwd_xcd(xr.ref(), xc.ref(), _lib.ptr(ar), _lib.ptr(ac), _lib.ptr(feat),
                                               ctx.negative_slope, ctx.p, ctx.seed, _lib.ptr(edge_max), _lib.ptr(edge_sum),
                                               _lib.ptr(o), _lib.ptr(g), _lib.ptr(grad_feat), _lib.ptr(grad_ar),
                                               _lib.ptr(grad_ac), _lib.ptr(ws), ws_bytes, v, n_src, h, fp, code,
                                               _lib.stream_of(g))
        if rc == _lib.EUNSUPPORTED:  # no plan, or a shape the plan kernels decline (column tiles): the ordinary entries
            ws, ws_bytes = _lib.workspace("cogdl_hip_gat_bwd_workspace_bytes", dev, v, n_src, h, fp, nnz, code)
            with _lib.on_device(dev):
                if ctx.p > 0.0:
                    rc = lib.cogdl_hip_gat_dropout_bwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(plan.colptr),
                                                       _lib.ptr(plan.rowind), _lib.ptr(plan.perm), _lib.ptr(ar),
                                                       _lib.ptr(ac), _lib.ptr(feat), ctx.negative_slope, ctx.p, ctx.seed,
                                                       _lib.ptr(edge_max), _lib.ptr(edge_sum), _lib.ptr(o), _lib.ptr(g),
                                                       _lib.ptr(grad_feat), _lib.ptr(grad_ar), _lib.ptr(grad_ac),
                                                       _lib.ptr(ws), ws_bytes, v, n_src, h, fp, nnz, code,
                                                       _lib.stream_of(g))
                else:
                    rc = lib.cogdl_hip_gat_bwd(_lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(plan.colptr),
                                               _lib.ptr(plan.rowind), _lib.ptr(ar), _lib.ptr(ac), _lib.ptr(feat),
                                               ctx.negative_slope, _lib.ptr(edge_max), _lib.ptr(edge_sum), _lib.ptr(o),
                                               _lib.ptr(g), _lib.ptr(grad_feat), _lib.ptr(grad_ar), _lib.ptr(grad_ac),
                                               _lib.ptr(ws), ws_bytes, v, n_src, h, fp, nnz, code, _lib.stream_of(g))
        _lib.check(rc, "gat_bwd")
        if fp != ctx.f:
            grad_feat = grad_feat[..., :ctx.f].contiguous()
        return (grad_ar.to(attn_row.dtype), grad_ac.to(attn_col.dtype), None, None, None, None, None,
                grad_feat.to(ctx.feat_dtype), None, None)


def fused_gat_func(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat):
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, negative_slope, in_feat)


def new_dropout_seed():
    """A 63-bit seed for one application of the fused attention dropout, drawn from torch's CPU generator: runs are
    reproducible under torch.manual_seed, and no device synchronisation is involved."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def fused_gat_dropout_func(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, p, seed=None):
    """`fused_gat_func` with nn.Dropout(p) on the attention (cogdl/layers/gat_layer.py:72-77 as one operator): the mask is
    a pure function of (seed, edge position, head); seed=None draws one from torch's generator."""
    if not 0.0 <= p <= 1.0:
        raise ValueError("dropout probability has to be between 0 and 1, but got %s" % p)
    if seed is None:
        if p > 0.0 and in_feat.is_cuda and torch.cuda.is_current_stream_capturing():
            # the seed is a kernel ARGUMENT: drawn here it would be frozen into the captured graph and every replay would
            # drop the same attention entries
            raise _lib.BackendError("fused_gat_dropout_func inside a hipGraph capture needs an explicit seed per replay "
                                    "(the seed is passed by value); capture with attn_drop = 0 or step eagerly")
        seed = new_dropout_seed() if p > 0.0 else 0
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, None, None, negative_slope, in_feat, float(p),
                                  int(seed))
