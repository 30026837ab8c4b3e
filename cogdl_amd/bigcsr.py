"""CSR graphs with 64-bit row pointers: 2^31 edges and more on one GPU.

The reference cannot run them: `spmm` casts `graph.row_indptr` to int32 (cogdl/utils/spmm_utils.py:106), the
CPU operator walks `int` offsets (cogdl/operators/spmm/spmm_cpu.cpp:24-33), the CUDA kernels take `int nnz`.
ogbn-papers100M as CogDL preprocesses it (symmetrised + coalesced, cogdl/datasets/ogb.py:50-55) has 3.2e9 edges and,
with 128-wide fp32 features, fits one 288 GB MI355X.

`csrspmm(rowptr, colind, x, csr_data)` (cogdl_amd/operators/spmm.py) takes this path when `rowptr` is int64: pass
`graph.row_indptr` itself instead of `.int()`.  The rows are cut into segments of ~2^29 edges, every segment is one
launch of the ordinary 32-bit kernels on rebased int32 row pointers (csrc/bigcsr.hip) -- results are those of the
32-bit operator row for row.  The plan (cuts, rebased row pointers, and for backward the 64-bit transpose) is cached
per (rowptr, colind) tensor identity: a caller of this path keeps its index tensors, it does not re-cast them per call.
"""
import ctypes
import os

import torch

from . import _lib
from .plan import tensor_key


# Row schedule of the segment launches (cogdl_hip_csr_spmm_i64_ordered): rows by decreasing degree inside windows of ROW_WINDOW
# rows.  The lane groups of a wave (two rows per wave at F = 128 fp32) and the waves of a workgroup then carry rows of one
# length -- on a power-law graph neighbouring ids differ by orders of magnitude, and a wave with an idle group keeps half its
# gathers in flight.  Windows keep the row pointer / output traffic of concurrently running workgroups together.  Measured on
# one eighth of the papers100M-shaped symmetrised graph (X = 7.1 GB; rows physically re-ordered, tools/exp/papers_order_ab.py,
# profiles/r06_papers_order_ab.txt): F = 128 fp32 42.8 -> 38.5 ms (0.635 -> 0.705 of 8 TB/s), whole-graph degree order 39.4 ms.
# At full size (3.2e9 edges, profiles/r06_papers_row_order.txt, r06_papers_sweep.txt): symmetrised forward 352 -> 292 ms; by window
# size 4096: 347 ms, 16384: 322, 65536: 292, 262144: 291, 2^20: 308, whole segment: 300; the long-row threshold (512) re-swept
# under the schedule: 256: 316 ms, 1024 / 2048: 293.
# Results are bit-identical (a row is still one lane group's sequential sum).  COGDL_AMD_ROW_ORDER=0 switches it off.
ORDER_ROWS = os.environ.get("COGDL_AMD_ROW_ORDER", "1") != "0"
ROW_WINDOW = int(os.environ.get("COGDL_AMD_ROW_WINDOW", 1 << 16))


def window_degree_order(rowptr, window=None):
    """int32 permutation of the rows of `rowptr` (any integer dtype, device tensor): decreasing degree inside windows of
    `window` rows (default ROW_WINDOW), windows in place."""
    window = ROW_WINDOW if window is None else int(window)
    m = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).long()
    top = int(deg.max()) + 1 if m else 1
    key = (torch.arange(m, device=rowptr.device) // window) * top + (top - 1 - deg)
    order = torch.argsort(key, stable=True)
    mode = os.environ.get("COGDL_AMD_ROW_SCHED", "desc")  # experiments: "asc", "mix" (tools/exp/papers_sweep2.sh)
    if mode == "asc":
        key = (torch.arange(m, device=rowptr.device) // window) * top + deg
        order = torch.argsort(key, stable=True)
    elif mode == "mix" and m >= 2 * window:
        # inside every full window: blocks of 8 rows taken alternately from the long and from the short half
        full = (m // window) * window
        o = order[:full].view(-1, window // 8, 8)
        half = o.shape[1] // 2
        mixed = torch.stack([o[:, :half], o[:, half:2 * half]], dim=2).reshape(o.shape[0], -1, 8)
        order = torch.cat([mixed.reshape(-1), order[full:]])
    return order.int()


class BigCsr:
    """Plan of one 64-bit CSR structure: segment cuts + rebased int32 row pointers + the row schedule (device)."""

    def __init__(self, rowptr, colind, n_cols=None, max_edges=0):
        dev = _lib.require_cuda(rowptr, colind)
        if rowptr.dtype != torch.int64 or colind.dtype != torch.int32:
            raise _lib.BackendError("64-bit CSR: rowptr must be int64 and colind int32 (got %s/%s)" % (rowptr.dtype, colind.dtype))
        if rowptr.dim() != 1 or colind.dim() != 1 or rowptr.numel() < 1:
            raise _lib.BackendError("rowptr/colind must be 1-D (rowptr non-empty)")
        self.rowptr, self.colind = rowptr.contiguous(), colind.contiguous()
        self.m, self.nnz = self.rowptr.numel() - 1, self.colind.numel()
        self.n_cols = self.m if n_cols is None else int(n_cols)
        self.seg = _lib.Segments()
        lib = _lib.hip()
        scratch = torch.empty(2 * (_lib.MAX_SEGMENTS + 1), dtype=torch.int64, device=dev)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_csr_segments(_lib.ptr(self.rowptr), self.m, self.nnz, int(max_edges), ctypes.addressof(self.seg),
                                            _lib.ptr(scratch), _lib.stream_of(self.rowptr))
            _lib.check(rc, "csr_segments")
            self.rowptr32 = torch.empty(self.m + self.seg.n, dtype=torch.int32, device=dev)
            rc = lib.cogdl_hip_csr_rebase_rowptr(_lib.ptr(self.rowptr), ctypes.addressof(self.seg), _lib.ptr(self.rowptr32),
                                                 _lib.stream_of(self.rowptr))
            _lib.check(rc, "csr_rebase_rowptr")
        self._seg_addr = ctypes.addressof(self.seg)
        self.row_order = self._row_schedule() if ORDER_ROWS and self.m > 0 and self.seg.n > 0 else None
        self._transposed = None
        self._val_key = self._val_src = self._val_t = None
        self._sym_checked = {}

    def _row_schedule(self):
        """[m] int32: per segment a permutation of its LOCAL row ids -- decreasing degree inside windows of ROW_WINDOW rows."""
        dev = self.rowptr.device
        order = torch.empty(self.m, dtype=torch.int32, device=dev)
        rows = self.segment_rows()
        for s in range(self.seg.n):
            r0, r1 = rows[s], rows[s + 1]
            if r1 <= r0:
                continue
            order[r0:r1] = window_degree_order(self.rowptr[r0:r1 + 1])
        return order

    @property
    def n_segments(self):
        return int(self.seg.n)

    def segment_rows(self):
        return [int(self.seg.row[i]) for i in range(self.seg.n + 1)]

    def segment_edges(self):
        return [int(self.seg.edge[i]) for i in range(self.seg.n + 1)]

    def nbytes(self):
        return 4 * self.rowptr32.numel() + (4 * self.row_order.numel() if self.row_order is not None else 0)

    # ---- operators ---------------------------------------------------------------------------------------------------
    def spmm(self, val, x, split_long_rows=True):
        """out = A x over all segments (cogdl_hip_csr_spmm_i64), on the current stream."""
        dev = _lib.require_cuda(self.rowptr32, val, x)
        if x.dim() != 2 or x.dtype not in _lib.DTYPE_CODE:
            raise _lib.BackendError("dense operand must be a 2-D f32/f16/bf16 tensor")
        x = x.contiguous()
        if val is not None:
            val = val.contiguous()
            if val.dtype != x.dtype:
                val = val.to(x.dtype)
            if val.numel() != self.nnz:
                raise _lib.BackendError("csr_data has %d entries for %d edges" % (val.numel(), self.nnz))
        k, code = x.shape[1], _lib.DTYPE_CODE[x.dtype]
        out = torch.empty((self.m, k), dtype=x.dtype, device=dev)
        lib = _lib.hip()
        ws, ws_bytes = None, 0
        if split_long_rows:
            ws_bytes = lib.cogdl_hip_csr_spmm_i64_workspace_bytes(self._seg_addr, k, code)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_csr_spmm_i64_ordered(_lib.ptr(self.rowptr32), self._seg_addr, _lib.ptr(self.colind), _lib.ptr(val),
                                                    _lib.ptr(x), _lib.ptr(out), k, code, _lib.ptr(self.row_order), _lib.ptr(ws),
                                                    ws_bytes, _lib.stream_of(x))
        _lib.check(rc, "csr_spmm_i64")
        return out

    def sddmm(self, d1, d2):
        """out[e] = <d1[row(e)], d2[col[e]]> (fp32) over all segments."""
        dev = _lib.require_cuda(self.rowptr32, d1, d2)
        d1, d2 = d1.contiguous().float(), d2.contiguous().float()
        out = torch.empty(self.nnz, dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            rc = _lib.hip().cogdl_hip_csr_sddmm_i64(_lib.ptr(self.rowptr32), self._seg_addr, _lib.ptr(self.colind), _lib.ptr(d1),
                                                    _lib.ptr(d2), _lib.ptr(out), d1.shape[1], _lib.stream_of(d1))
        _lib.check(rc, "csr_sddmm_i64")
        return out

    def transpose(self, val=None, keep_perm=True):
        """Stable transpose -> (BigCsr of A^T, perm64 | None, val_t | None).  `val` given: val_t = val[perm] is produced
        inside the transpose (no 8-byte-per-edge perm needed when keep_perm is False)."""
        dev = self.rowptr32.device
        lib = _lib.hip()
        colptr = torch.empty(self.n_cols + 1, dtype=torch.int64, device=dev)
        rowind = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        perm = torch.empty(self.nnz, dtype=torch.int64, device=dev) if keep_perm else None
        val_t, vb = None, 0
        if val is not None:
            val = val.contiguous()
            vb = val.element_size()
            if vb not in (2, 4) or val.numel() != self.nnz:
                raise _lib.BackendError("transpose: values must be [nnz] with 2- or 4-byte elements")
            val_t = torch.empty_like(val)
        ws_bytes = lib.cogdl_hip_csr2csc_i64_workspace_bytes(self._seg_addr, self.n_cols)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_csr2csc_i64(_lib.ptr(self.rowptr32), self._seg_addr, _lib.ptr(self.colind), self.n_cols,
                                           _lib.ptr(colptr), _lib.ptr(rowind), _lib.ptr(perm), _lib.ptr(val), _lib.ptr(val_t), vb,
                                           _lib.ptr(ws), ws_bytes, _lib.stream_of(colptr))
        _lib.check(rc, "csr2csc_i64")
        del ws
        return BigCsr(colptr, rowind, n_cols=self.m), perm, val_t

    # ---- the caller's `sym` flag ---------------------------------------------------------------------------------------
    def symmetric_verified(self, w, probes=4, tol=1e-5):
        """True when A (with the values `w`) passes a randomised symmetry test: for seeded random X, Y in R^{n x probes},
        <Y, A X> = <X, A Y> column by column iff (A - A^T) is orthogonal to X Y^T - Y X^T -- an asymmetric A fails with a
        difference of the order ||A - A^T||_F, a symmetric one passes up to fp32 rounding (~1e-7 ||A||_F).  The threshold is
        `tol` * ||A||_F.  Two forward passes at width `probes`, once per (structure, weight tensor); the answer is cached.

        Why: the reference's SPMMFunction trusts `sym` blindly and multiplies by A again in backward
        (cogdl/operators/spmm.py:63-66) -- wrong for row-normalised weights on a symmetric STRUCTURE.  Here the flag only
        saves the transpose (26 GB of indices and values at papers100M size) after the matrix has passed this test."""
        if self.m != self.n_cols:
            return False
        key = None if w is None else tensor_key(w)
        hit = self._sym_checked.get(key)
        if hit is not None and (w is None or hit[1] is w):
            return hit[0]
        dev = self.rowptr32.device
        g = torch.Generator(device=dev)
        g.manual_seed(0x5EED)
        x = torch.randn(self.m, probes, device=dev, generator=g)
        y = torch.randn(self.m, probes, device=dev, generator=g)
        wv = None if w is None else w.detach().float()
        s1 = (y.double() * self.spmm(wv, x).double()).sum(0)
        s2 = (x.double() * self.spmm(wv, y).double()).sum(0)
        fro = float(self.nnz) ** 0.5 if wv is None else float(torch.linalg.vector_norm(wv))
        ok = bool(((s1 - s2).abs().max() <= tol * max(fro, 1e-30)).item())
        if len(self._sym_checked) >= 4:
            self._sym_checked.pop(next(iter(self._sym_checked)))
        self._sym_checked[key] = (ok, w)
        return ok

    # ---- cached transpose for autograd -------------------------------------------------------------------------------
    def transposed(self, w):
        """(BigCsr of A^T, w[perm] | None) for the backward pass; both cached (constant weights are moved once, keyed on
        the weight tensor's identity as CscPlan.transposed_values does).  The first call with constant weights fuses
        their permutation into the transpose and keeps no perm (8 bytes per edge); weights that take part in autograd
        need it and get it."""
        need_perm = w is not None and w.requires_grad
        if self._transposed is None or (need_perm and self._transposed[1] is None):
            fuse = w is not None and not need_perm
            t, perm, val_t = self.transpose(w.detach() if fuse else None, keep_perm=need_perm)
            self._transposed = (t, perm)
            if fuse:
                self._val_key, self._val_src, self._val_t = tensor_key(w), w.detach(), val_t
        t, perm = self._transposed
        if w is None:
            return t, None
        if need_perm:
            return t, gather_rows_i64(perm, w.detach())
        key = tensor_key(w)
        if key != self._val_key or self._val_src is None:
            if perm is None:
                # a SECOND constant weight tensor on this structure (a caller that re-creates its weights per call, e.g. the
                # dispatcher's `csr_data.half()` under fp16): build the permutation once and gather from now on, instead
                # of transposing the structure again for every new tensor
                t, perm, _ = self.transpose(None, keep_perm=True)
                self._transposed = (t, perm)
            val_t = gather_rows_i64(perm, w.detach())
            self._val_key, self._val_src, self._val_t = key, w.detach(), val_t
        return t, self._val_t


def gather_rows_i64(perm, src):
    """out[i] = src[perm[i]] along dim 0 with 64-bit positions."""
    dev = _lib.require_cuda(perm, src)
    src = src.contiguous()
    out = torch.empty_like(src)
    n = perm.numel()
    h = src.numel() // max(n, 1) if n else 0
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_gather_rows_i64(_lib.ptr(perm), _lib.ptr(src), _lib.ptr(out), n, h, src.element_size(),
                                                  _lib.stream_of(src))
    _lib.check(rc, "gather_rows_i64")
    return out


# plans by tensor identity (the key holds references to both tensors: their addresses cannot be recycled while it lives)
_BIG_PLANS = {}
_BIG_PLANS_MAX = 4  # (a plan pins its graph's index tensors: clear_plans() releases them)


def plan_of(rowptr, colind, n_cols):
    key = (tensor_key(rowptr), tensor_key(colind), int(n_cols))
    hit = _BIG_PLANS.get(key)
    if hit is not None:
        return hit[0]
    if len(_BIG_PLANS) >= _BIG_PLANS_MAX:
        _BIG_PLANS.pop(next(iter(_BIG_PLANS)))
    plan = BigCsr(rowptr, colind, n_cols=n_cols)
    _BIG_PLANS[key] = (plan, rowptr, colind)
    return plan


def clear_plans():
    _BIG_PLANS.clear()


class BigSPMMFunction(torch.autograd.Function):
    """SPMMFunction (cogdl/operators/spmm.py:43-80) for a 64-bit CSR: forward = the segmented csr_spmm, backward =
    the same on the cached 64-bit transpose (grad_x) and the segmented sddmm (grad of the edge weights)."""

    @staticmethod
    def forward(ctx, rowptr, colind, feat, edge_weight_csr=None, sym=False):
        plan = plan_of(rowptr, colind, feat.shape[0])
        out = plan.spmm(edge_weight_csr, feat)
        ctx.plan = plan
        ctx.sym = bool(sym)
        need_w = edge_weight_csr is not None and ctx.needs_input_grad[3]
        ctx.save_for_backward(edge_weight_csr, feat if need_w else None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        w, feat = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_feat = grad_w = None
        if ctx.needs_input_grad[2]:
            # `sym` (Graph.is_symmetric(), passed on by the dispatcher): A^T = A, the forward plan serves the backward --
            # but only after the matrix has passed the symmetry test (the reference's blind trust is its `sym` bug)
            if ctx.sym and ctx.plan.symmetric_verified(w):
                t, w_t = ctx.plan, (None if w is None else w.detach())
            else:
                t, w_t = ctx.plan.transposed(w)
            grad_feat = t.spmm(w_t, grad_out)
        if w is not None and ctx.needs_input_grad[3]:
            grad_w = ctx.plan.sddmm(grad_out, feat.detach()).to(w.dtype)
        return None, None, grad_feat, grad_w, None
