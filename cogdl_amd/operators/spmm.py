"""Drop-in for cogdl/operators/spmm.py: exports `csrspmm` and `spmm_cpu` with the reference's
call signatures (operators/spmm.py:24-27,35-40), backed by libcogdl_hip / libcogdl_host.

    csrspmm(rowptr:int32[M+1], colind:int32[nnz], x:[N,F], csr_data:[nnz]|None, sym=False, actnn=False) -> [M,F]
    spmm_cpu(rowptr, colind, csr_data, x) -> [M,F]          (CPU tensors; note the argument order)

Differences from the reference, all deliberate (SURVEY.md section 7 "hard parts"):
  * backward always uses the true transpose A^T (cached per structure, cogdl_amd/plan.py).  The
    reference reuses A when `sym` is set (operators/spmm.py:61-62), which is only right for a
    truly symmetric weighted graph -- for those both give the same operator.
  * unweighted + not sym no longer calls csr2csc(None) (operators/spmm.py:75).
  * failures raise BackendError instead of silently falling back to torch.scatter_add.
  * actnn=True (activation quantisation, third_party/actnn absent) is rejected loudly.
  * an int64 `rowptr` (with int32 `colind`) selects the 64-bit CSR path (cogdl_amd/bigcsr.py): graphs of 2^31 edges
    and more, which the reference's `.int()` cast cannot express.
"""
import os

import torch

from .. import _lib
from .. import plan as _plan
from .. import xcdplan
from ..plan import PLANS, Fingerprint, fingerprint_of, csr2csc, gather_rows

_lib.hip()  # fail at import if the HIP library is missing (no silent `csrspmm = None`)

# bench.py sets this to a list to get (start, end) HIP events around every csr_spmm kernel launch,
# recorded on the stream the kernel is launched on.
KERNEL_EVENTS = None


class KernelEventLog(list):
    """A KERNEL_EVENTS list that brackets every `every`-th launch only.  Recording a timing event costs the HOST ~18 us on this
    stack (tools/headline_host_profile.py: the dispatcher call + backward on a 2048-row graph 174 -> 247 us per step with the four
    records of a step; creating the events ahead of time changes nothing -- it is the record): bracketing every launch of a
    timed loop of two 0.18 ms kernels makes the loop's wall time depend on the host (one box of the pool: 0.46 instead of 0.39 ms
    per step with identical kernel times).  The sampled launches are inside the timed region all the same."""

    def __init__(self, every=1):
        super().__init__()
        self.every, self.seen, self.enabled = max(1, int(every)), 0, True

    def take(self):
        if not self.enabled:
            return None
        self.seen += 1
        if (self.seen - 1) % self.every:
            return None
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _check_csr(rowptr, colind, x):
    if rowptr.dtype != torch.int32 or colind.dtype != torch.int32:
        raise _lib.BackendError("rowptr/colind must be int32 (got %s/%s)" % (rowptr.dtype, colind.dtype))
    if x.dim() != 2:
        raise _lib.BackendError("dense operand must be 2-D, got shape %s" % (tuple(x.shape),))
    if x.dtype not in _lib.DTYPE_CODE:
        raise _lib.BackendError("unsupported dtype %s" % x.dtype)


def csr_spmm_raw(rowptr, colind, val, x, variant=-1, out=None, split_long_rows=True, row_order=None):
    """One cogdl_hip_csr_spmm launch on the current stream (no autograd).  With `out` given the result is
    accumulated into it (out += A x, cogdl_hip_csr_spmm_acc).  row_order: an int32 permutation of the rows, the row blocks'
    schedule (cogdl_hip_csr_spmm_ordered; same results bit for bit)."""
    dev = _lib.require_cuda(rowptr, colind, val, x)
    _check_csr(rowptr, colind, x)
    x = x.contiguous()
    rowptr, colind = rowptr.contiguous(), colind.contiguous()
    if val is not None:
        val = val.contiguous()
        if val.dtype != x.dtype:
            val = val.to(x.dtype)
        if val.numel() != colind.numel():
            raise _lib.BackendError("csr_data has %d entries for %d edges" % (val.numel(), colind.numel()))
    m, k, nnz = rowptr.numel() - 1, x.shape[1], colind.numel()
    acc = out is not None
    if acc:
        if out.shape != (m, k) or out.dtype != x.dtype or not out.is_contiguous():
            raise _lib.BackendError("accumulation target must be a contiguous [%d, %d] %s tensor" % (m, k, x.dtype))
    else:
        out = torch.empty((m, k), dtype=x.dtype, device=dev)
    lib = _lib.hip()
    code = _lib.DTYPE_CODE[x.dtype]
    ws, ws_bytes = (_lib.workspace("cogdl_hip_csr_spmm_workspace_bytes", dev, nnz, k, code) if split_long_rows
                    else (None, 0))
    with _lib.on_device(dev):
        pair = None
        if KERNEL_EVENTS is not None:
            pair = (KERNEL_EVENTS.take() if hasattr(KERNEL_EVENTS, "take")
                    else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        if pair is not None:
            ev0, ev1 = pair
            ev0.record()
        if row_order is not None:
            if row_order.dtype != torch.int32 or row_order.numel() != m or not row_order.is_contiguous():
                raise _lib.BackendError("row_order must be a contiguous int32 permutation of the %d rows" % m)
            rc = lib.cogdl_hip_csr_spmm_ordered(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(val), _lib.ptr(x), _lib.ptr(out), m, k,
                                                nnz, code, 1 if acc else 0, _lib.ptr(row_order), _lib.ptr(ws), ws_bytes,
                                                _lib.stream_of(x))
        elif acc:
            rc = lib.cogdl_hip_csr_spmm_acc(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(val), _lib.ptr(x),
                                            _lib.ptr(out), m, k, nnz, code, _lib.ptr(ws), ws_bytes,
                                            _lib.stream_of(x))
        else:
            rc = lib.cogdl_hip_csr_spmm_variant(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(val), _lib.ptr(x),
                                                _lib.ptr(out), m, k, nnz, code, variant, _lib.ptr(ws), ws_bytes,
                                                _lib.stream_of(x))
        if pair is not None:
            ev1.record()
            KERNEL_EVENTS.append((ev0, ev1))
    _lib.check(rc, "csr_spmm")
    return out


def csr_spmm_xcd_raw(xplan, val, x, out=None):
    """A x over an XCD-partitioned plan of the structure (cogdl_amd/xcdplan.py; cogdl_hip_csr_spmm_xcd): `val` in the
    CALLER's edge order (the plan's `eid` says which entry every plan position takes), rows of more than xcdplan.SPLIT edges
    re-associated.  With `out` given: out += A x."""
    dev = _lib.require_cuda(val, x)
    if x.dim() != 2 or x.dtype not in _lib.DTYPE_CODE:
        raise _lib.BackendError("dense operand must be a 2-D float32/float16/bfloat16 tensor")
    x = x.contiguous()
    if val is not None:
        if val.numel() != xplan.nnz:
            raise _lib.BackendError("csr_data has %d entries for %d edges" % (val.numel(), xplan.nnz))
        val = xplan.permuted_values(val.contiguous() if val.dtype == x.dtype else val.to(x.dtype))
    m, k = xplan.m, x.shape[1]
    acc = out is not None
    if acc:
        if out.shape != (m, k) or out.dtype != x.dtype or not out.is_contiguous():
            raise _lib.BackendError("accumulation target must be a contiguous [%d, %d] %s tensor" % (m, k, x.dtype))
    else:
        out = torch.empty((m, k), dtype=x.dtype, device=dev)
    code = _lib.DTYPE_CODE[x.dtype]
    ws, ws_bytes = _lib.workspace("cogdl_hip_csr_spmm_xcd_workspace_bytes", dev, xplan.n_parts, k, code)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_csr_spmm_xcd(xplan.ref(), _lib.ptr(val), _lib.ptr(x), _lib.ptr(out), m, k, code,
                                               1 if acc else 0, _lib.ptr(ws), ws_bytes, _lib.stream_of(x))
    _lib.check(rc, "csr_spmm_xcd")
    return out


SPLIT_16 = 256  # (= xcdplan.SPLIT: 16-bit csr_spmm and the fused GAT operator share one plan per structure)


def _xcd_split(rowptr, colind, x, fp=None):
    """Does this csr_spmm launch take an XCD-partitioned plan (cogdl_amd/xcdplan.py), and cut at which row length?
    -> None (no plan) or the split.
      fp32    hub-heavy structures over cache-sized tables (xcdplan.wanted): yes, with split = the exact-row bound of the
              ordinary path (cogdl_hip_exact_row_edges(nnz)) -- rows up to that length stay whole and bit-identical to the
              reference loop, as before; only the rows the ordinary path already re-associates (its long-row pieces) are cut by
              owner XCD instead of into contiguous chunks.  Measured on the Reddit-shaped graph, F = 64: see
              profiles/r06_xcd_spmm_fp32.txt.
      16-bit  the same structures and tables, cut at SPLIT_16 = 256 edges (no bit-exact contract to keep).  Measured on the
              Reddit-shaped graph (profiles/r06_xcd_spmm_split.txt): bf16 F = 64 1524 -> 1316 us, F = 128 3241 -> 2291 us; at
              split 64 / 1024: 1386 / 1345 us.  (Before the hub rows' part records were merged by whole workgroups -- rowreduce.h:
              rowreduce_vcombine_kernel -- the same plan LOST: 1512 -> 1638 us, profiles/r06_xcd_quick.txt.)"""
    if x.dim() != 2 or x.dtype not in _lib.DTYPE_CODE:
        return None
    m, nnz = rowptr.numel() - 1, colind.numel()
    if xcdplan.MODE == "force":
        return xcdplan.SPLIT if xcdplan.wanted(m, nnz, x.shape[0], x.shape[1] * x.element_size()) else None
    if xcdplan.wanted(m, nnz, x.shape[0], x.shape[1] * x.element_size()):
        return int(_lib.hip().cogdl_hip_exact_row_edges(nnz)) if x.dtype == torch.float32 else SPLIT_16
    # skewed structures of any size, when the structure's fingerprint is on the host already (xcdplan.ordered_wanted): cut at the
    # exact-row bound whatever the dtype (rows up to it stay whole and sequential, as in the ordinary launch)
    if xcdplan.ordered_wanted(fp, rowptr, m, nnz, x.shape[0], x.shape[1] * x.element_size()):
        return int(_lib.hip().cogdl_hip_exact_row_edges(nnz))
    return None


def csr_sddmm_raw(rowptr, colind, d1, d2):
    """out[e] = <d1[row(e)], d2[col[e]]>  (fp32)."""
    dev = _lib.require_cuda(rowptr, colind, d1, d2)
    d1, d2 = d1.contiguous().float(), d2.contiguous().float()
    m, k = rowptr.numel() - 1, d1.shape[1]
    nnz = colind.numel()
    out = torch.empty(nnz, dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_csr_sddmm(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(d1), _lib.ptr(d2),
                                            _lib.ptr(out), m, k, nnz, _lib.stream_of(d1))
    _lib.check(rc, "csr_sddmm")
    return out


class SPMMFunction(torch.autograd.Function):
    """Mirrors cogdl.operators.spmm.SPMMFunction (operators/spmm.py:43-80)."""

    @staticmethod
    def forward(ctx, rowptr, colind, feat, edge_weight_csr=None, sym=False):
        # The structure hash is enqueued BEFORE the SpMM so that it has landed in pinned host memory long
        # before backward asks for it (Fingerprint.key() waits on an event recorded right behind the hash
        # kernel, not behind the SpMM).
        rowptr, colind = _lib.csr_structure(rowptr, colind)  # validated + contiguous before anything reads raw pointers
        ctx.transient = _plan.transient()  # (the dense operand is checked by csr_spmm_raw)
        taped = _plan._TAPE is not None and not ctx.transient
        ctx.xcd = None if taped else _xcd_split(rowptr, colind, feat)
        memo = getattr(rowptr, "_cogdl_amd_struct", None) is not None
        if ctx.transient:
            ctx.fp = None
        elif ctx.needs_input_grad[2] or ctx.xcd is not None or memo or taped:
            ctx.fp = fingerprint_of(rowptr, colind, feat.shape[0])
        else:
            ctx.fp = _plan.known_fingerprint(rowptr, colind, feat.shape[0])  # (never hashes: inference calls stay as they were)
        xplan = None
        if taped:
            # cogdl_amd.graphs.capture: the recorded eager run waits for the structure's key and decides as a call with a known
            # fingerprint would; the capture takes the recorded decision (nothing is hashed or read back while capturing)
            def decide():
                ctx.fp.key()
                split = _xcd_split(rowptr, colind, feat, ctx.fp)
                return split, (xcdplan.csr_plan(ctx.fp, rowptr, colind, split) if split is not None else None)

            ctx.xcd, xplan = _plan.taped_choice("csr_spmm.forward", decide)
        elif ctx.xcd is None and ctx.fp is not None:
            # a memoised fingerprint (install(structure_memo=True)): its key costs ONE wait per structure, after which skewed
            # structures of any size take a plan -- deterministically, from the first call on.  The identity memo (the same
            # index tensor objects as an earlier call, plan.fingerprint_of) has its key once a backward pass has asked for it.
            if memo and ctx.fp.event is not None:
                ctx.fp.key()
            if ctx.fp._key is not None:
                ctx.xcd = _xcd_split(rowptr, colind, feat, ctx.fp)
        if ctx.xcd is not None:
            if xplan is None:
                xplan = xcdplan.csr_plan(ctx.fp, rowptr, colind, ctx.xcd)
            out = csr_spmm_xcd_raw(xplan, edge_weight_csr, feat)
        else:
            out = csr_spmm_raw(rowptr, colind, edge_weight_csr, feat)
        need_w = edge_weight_csr is not None and ctx.needs_input_grad[3]
        ctx.n_src = feat.shape[0]
        ctx.sym = sym
        ctx.save_for_backward(rowptr, colind, edge_weight_csr, feat if need_w else None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, colind, w, feat = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_feat = grad_w = None
        if ctx.needs_input_grad[2]:
            if ctx.transient:  # (plan.transient_structures: a one-off structure, transposed here, nothing cached or read back)
                plan = csr2csc(rowptr, colind, ctx.n_src, padded=True)
                w_t = gather_rows(plan.perm, w.detach()) if w is not None else None
                grad_feat = csr_spmm_raw(plan.colptr, plan.rowind, w_t, grad_out)
            else:
                plan = PLANS.get(ctx.fp, rowptr, colind, ctx.n_src)
                # (the key is known here -- PLANS.get has waited for the hash: the transpose of a skewed structure takes a plan
                #  whether or not the forward call could -- from its SECOND sighting on: a structure that never comes back, a
                #  sampled block passed without plan.transient_structures(), must not pay a plan build of milliseconds)

                def decide():
                    split_t = _xcd_split(plan.colptr, plan.rowind, grad_out, ctx.fp if (ctx.xcd is not None or getattr(plan, "sightings", 1) > 1) else None)
                    return split_t, (xcdplan.csc_plan(ctx.fp, plan, split_t) if split_t is not None else None)

                split_t, xplan_t = _plan.taped_choice("csr_spmm.backward", decide)
                if split_t is not None:
                    # (w stays in CSR order: the plan of the transpose maps its positions through the transpose's perm)
                    grad_feat = csr_spmm_xcd_raw(xplan_t, w, grad_out)
                else:
                    w_t = plan.transposed_values(w) if w is not None else None
                    grad_feat = csr_spmm_raw(plan.colptr, plan.rowind, w_t, grad_out, split_long_rows=plan.has_hub_columns())
        if w is not None and ctx.needs_input_grad[3]:
            grad_w = csr_sddmm_raw(rowptr, colind, grad_out, feat.detach()).to(w.dtype)
        return None, None, grad_feat, grad_w, None


def csr_spmm_epilogue_raw(rowptr, colind, val, x, src_scale=None, dst_scale=None, bias=None, relu=False,
                          split_long_rows=True):
    """One cogdl_hip_csr_spmm_epilogue launch: relu?( dst_scale * (A (src_scale * x)) + bias ), fp32, no autograd."""
    dev = _lib.require_cuda(rowptr, colind, val, x, src_scale, dst_scale, bias)
    _check_csr(rowptr, colind, x)
    if x.dtype != torch.float32:
        raise _lib.BackendError("the fused epilogue is fp32 (got %s)" % x.dtype)
    x, rowptr, colind = x.contiguous(), rowptr.contiguous(), colind.contiguous()
    m, k, nnz = rowptr.numel() - 1, x.shape[1], colind.numel()

    def vec(t, n, what):
        if t is None:
            return None
        t = t.reshape(-1).float().contiguous()
        if t.numel() != n:
            raise _lib.BackendError("%s has %d entries, expected %d" % (what, t.numel(), n))
        return t

    val = vec(val, nnz, "csr_data")
    src_scale, dst_scale, bias = vec(src_scale, x.shape[0], "out_norm"), vec(dst_scale, m, "in_norm"), vec(bias, k, "bias")
    out = torch.empty((m, k), dtype=torch.float32, device=dev)
    ws, ws_bytes = (_lib.workspace("cogdl_hip_csr_spmm_workspace_bytes", dev, nnz, k, 0) if split_long_rows else (None, 0))
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_csr_spmm_epilogue(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(val), _lib.ptr(x),
                                                    _lib.ptr(out), m, k, nnz, 0, _lib.ptr(src_scale), _lib.ptr(dst_scale),
                                                    _lib.ptr(bias), 1 if relu else 0, _lib.ptr(ws), ws_bytes,
                                                    _lib.stream_of(x))
    _lib.check(rc, "csr_spmm_epilogue")
    return out


class FusedSPMMFunction(torch.autograd.Function):
    """y = relu?( in_norm * (A (out_norm * x)) + bias ) in one kernel (forward) and
    grad_x = out_norm * (A^T (in_norm * (grad_y * [y > 0]))) in one kernel (backward: the same epilogue kernel on the
    cached transpose with the two norm vectors swapped).  Norms and edge weights are constants of the graph (no
    gradient), as in CogDL's dispatcher (cogdl/utils/spmm_utils.py:98-109)."""

    @staticmethod
    def forward(ctx, rowptr, colind, feat, edge_weight_csr, out_norm, in_norm, bias, relu, transient=False,
                max_row_edges=None):
        rowptr, colind = _lib.csr_structure(rowptr, colind)
        _check_csr(rowptr, colind, feat)
        ctx.transient = bool(transient) or _plan.transient()
        ctx.fp = fingerprint_of(rowptr, colind, feat.shape[0]) if ctx.needs_input_grad[2] and not ctx.transient else None
        # a caller that KNOWS no row reaches the long-row threshold (a sampled block: at most `fanout` edges per row)
        # saves the launch of the long-row combine kernel; the result is the same either way (no such row is split)
        split = max_row_edges is None or int(max_row_edges) > _lib.hip().cogdl_hip_long_row_threshold(colind.numel())
        out = csr_spmm_epilogue_raw(rowptr, colind, edge_weight_csr, feat, out_norm, in_norm, bias, relu,
                                    split_long_rows=split)
        side = _plan.early_transpose_stream() if ctx.transient and ctx.needs_input_grad[2] else None
        # (plan.transient_structures(side_stream=...): the transpose the backward will need starts NOW, on the side stream)
        ctx.early_plan = csr2csc(rowptr, colind, feat.shape[0], padded=True, stream=side) if side is not None else None
        ctx.n_src, ctx.relu, ctx.has_bias = feat.shape[0], bool(relu), bias is not None
        ctx.save_for_backward(rowptr, colind, edge_weight_csr, out_norm, in_norm, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, colind, w, out_norm, in_norm, out = ctx.saved_tensors
        g = grad_out.contiguous()
        if ctx.relu:
            g = g * (out > 0).to(g.dtype)
        grad_feat = grad_bias = None
        if ctx.has_bias and ctx.needs_input_grad[6]:
            grad_bias = g.sum(0)
        if ctx.needs_input_grad[2]:
            if ctx.transient and ctx.early_plan is not None:
                plan = ctx.early_plan
                if plan.ready is not None:  # (None: a block without rows -- nothing was launched)
                    torch.cuda.current_stream(g.device).wait_event(plan.ready)
                w_t = gather_rows(plan.perm, w.detach()) if w is not None else None
                hubs = True
            elif ctx.transient:  # a structure that is never seen again: transposed here, nothing hashed or cached or read back
                plan = csr2csc(rowptr, colind, ctx.n_src, padded=True)
                w_t = gather_rows(plan.perm, w.detach()) if w is not None else None
                hubs = True
            else:
                plan = PLANS.get(ctx.fp, rowptr, colind, ctx.n_src)
                w_t = plan.transposed_values(w) if w is not None else None
                hubs = plan.has_hub_columns()
            grad_feat = csr_spmm_epilogue_raw(plan.colptr, plan.rowind, w_t, g, in_norm, out_norm, None, False,
                                              split_long_rows=hubs)
        return None, None, grad_feat, None, None, None, grad_bias, None, None, None


def csrspmm_fused(rowptr, colind, x, csr_data=None, out_norm=None, in_norm=None, bias=None, relu=False):
    """relu?( in_norm * csrspmm(rowptr, colind, out_norm * x, csr_data) + bias ) as one operator (fp32).  out_norm:
    [N_src] or [N_src, 1], in_norm: [M] or [M, 1] (Graph.out_norm / Graph.in_norm, cogdl/data/data.py:240-258)."""
    return FusedSPMMFunction.apply(rowptr, colind, x, csr_data, out_norm, in_norm, bias, relu)


def csrspmm_block(rowptr, colind, x, csr_data=None, in_norm=None, max_row_edges=None):
    """in_norm * (A x) for a SAMPLED block (fp32): a structure that changes with every mini-batch, so the backward
    transposes it on the spot instead of hashing it into the plan cache (no structure hash, no pinned read-back, no
    host synchronisation anywhere -- the call can be captured in a hipGraph, and the transposes of a million
    mini-batches do not pile up in the cache).  `rowptr` may describe fewer edges than `colind` holds
    (rowptr[-1] <= len(colind): the fixed-capacity blocks of sample_adj_padded); the surplus entries are ignored in both
    directions (they only cost time).
    in_norm = 1 / in-degree gives the mean aggregator (Graph.row_norm, cogdl/data/data.py:240-258).
    max_row_edges: an upper bound of the edges per row when the caller knows one (default: what the sampler attached to
    `rowptr` -- its fan-out; None = unknown): below the long-row threshold the forward is one launch instead of two."""
    if max_row_edges is None:
        max_row_edges = getattr(rowptr, "_cogdl_max_row_edges", None)
    return FusedSPMMFunction.apply(rowptr, colind, x, csr_data, None, in_norm, None, False, True, max_row_edges)


def csrspmm(rowptr, colind, x, csr_data, sym=False, actnn=False):
    if actnn:
        raise _lib.BackendError("actnn=True needs the ActNN quantiser (third_party/actnn is an empty submodule "
                                "in the reference); not supported by the HIP backend")
    if rowptr.dtype == torch.int64:
        # 64-bit CSR (cogdl_amd/bigcsr.py): `graph.row_indptr` itself instead of `.int()` -- the only way to express a
        # graph of 2^31 edges or more (the reference's cast wraps, cogdl/utils/spmm_utils.py:106)
        from ..bigcsr import BigSPMMFunction

        return BigSPMMFunction.apply(rowptr, colind, x, csr_data, sym)
    return SPMMFunction.apply(rowptr, colind, x, csr_data, sym)


def spmm_cpu(rowptr, colind, csr_data, x):
    """CogDL's CPU SpMM (csr_spmm_cpu, operators/spmm/spmm_cpu.cpp:39-58) on host threads."""
    for name, t in (("rowptr", rowptr), ("colind", colind), ("csr_data", csr_data), ("x", x)):
        if t is not None and t.device.type != "cpu":
            raise _lib.BackendError("spmm_cpu: %s must be a CPU tensor" % name)
    if rowptr.dtype not in (torch.int32, torch.int64) or colind.dtype != torch.int32 or x.dtype != torch.float32:
        raise _lib.BackendError("spmm_cpu expects int32 (or int64 row pointer) indices and float32 features")
    rowptr, colind, x = rowptr.contiguous(), colind.contiguous(), x.contiguous()
    val = None if csr_data is None else csr_data.contiguous().float()
    m, k = rowptr.numel() - 1, x.shape[1]
    out = torch.empty((m, k), dtype=torch.float32)
    nthreads = int(os.environ.get("COGDL_AMD_CPU_THREADS", torch.get_num_threads()))
    # (an int64 row pointer = a graph of 2^31 edges and more, which the reference's `int` loop cannot walk)
    fn = _lib.host().cogdl_host_csr_spmm_f32 if rowptr.dtype == torch.int32 else _lib.host().cogdl_host_csr_spmm_f32_i64
    rc = fn(_lib.ptr(rowptr), _lib.ptr(colind), _lib.ptr(val), _lib.ptr(x), _lib.ptr(out), m, k, nthreads)
    _lib.check_host(rc, "csr_spmm_cpu")
    return out
