"""Drop-in for cogdl/operators/ops.py: `scatter_add`, `op_aggr` and the `s_*` message operators, same names, arguments
and results (reference lines cited per function).

The reference composes everything from torch ops over the COO edge list: the aggregating operators materialise the
[E, F] message tensor and `scatter_add_` it (ops.py:4-11, 43-52; atomics on a GPU, so the fp32 sums depend on the run).
Here fp32 GPU inputs of the aggregating operators (`scatter_add`, `op_aggr`, `s_{add,sub,mul}_e_{sum,mean}`) go to one
fused HIP kernel (cogdl_hip_gspmm, csrc/gspmm.hip) over the destination-sorted view of the edges: no message tensor,
no atomics, per output element the edges are added in the caller's edge order.  The destination sort is a stable GPU
sort done once per edge list (cogdl_hip_coo2csr_index) and memoised on the identity of the index tensors; the caller's
tensors and the Graph are never reordered.  Everything else -- CPU tensors, other dtypes, and the purely elementwise
`s_*_e` / `s_*_t` operators (a gather and one arithmetic op: torch already runs those at memory speed) -- executes the
reference's own torch expressions.  A GPU call of an AGGREGATING operator that lands there (half precision, a width the
kernel does not broadcast, a 2-D index) says so once per (operator, reason) with a `TorchRouteWarning`: these operators
ARE torch compositions in the reference, so the route is the reference's semantics, not a fallback from a failed kernel
-- but it is never silent.
"""
import collections

import torch

from .. import _lib
from ..plan import tensor_key

_lib.hip()

_OPS = {"add": 0, "sub": 1, "mul": 2}  # COGDL_HIP_GSPMM_*
_OPS_WMUL = 3  # COGDL_HIP_GSPMM_WMUL: (x * weight) * efeat, autograd's rounding order in the backward of "mul"


# --------------------------------------------------------------------------------------------- destination plans
class EdgePlan:
    """Destination-sorted view of an edge list: rowptr int32 [n+1], perm int32 [E] (sorted position -> edge id; stable),
    sorted (perm is the identity)."""
    __slots__ = ("rowptr", "perm", "sorted", "n", "keep", "_col_key", "_col_src", "_colind", "uses", "_skewed", "_xcd_key", "_xcd", "_xcd_builds")

    def __init__(self, dst, n):
        from ..graph_build import coo2csr_index

        row_ptr, perm = coo2csr_index(dst, dst, n)  # raises on ids outside [0, n) (scatter_add_ would, too)
        self.rowptr = row_ptr.int()
        self.perm = perm.int()
        e = dst.numel()
        self.sorted = bool(e == 0 or (perm == torch.arange(e, device=dst.device)).all().item())
        self.n = n
        self.keep = dst  # pins the key tensor: its data_ptr cannot be recycled while the plan lives
        self._col_key, self._col_src, self._colind = None, None, None
        self.uses, self._skewed, self._xcd_key, self._xcd, self._xcd_builds = 0, None, None, None, 0

    XCD_MIN_COLUMNS = 65  # (narrower rows: the plan's launch measured 8-15 % SLOWER than the ordinary one, below)

    def xcd(self, colind, k):
        """The XCD-partitioned, length-ordered plan of this view (cogdl_amd/xcdplan.py) for the launch that walks `colind` (int32,
        sorted order), or None.  Taken by memoised edge lists of SKEWED graphs from their second use on (the plan build is a
        few torch sorts: a one-off edge list must not pay it), cut at the exact-row bound of the ordinary launch -- rows up to it
        are summed in the caller's edge order as before, longer rows in pieces as before.  Measured on the arxiv-sized R-MAT
        graph, shuffled COO list, ordinary launch -> plan (profiles/r06_gspmm_plan_ab.txt): s_mul_e_sum F = 128 747 -> 460 us,
        F = 96 497 -> 372, F = 256 1432 -> 917; scatter_add F = 128 622 -> 383; forward + backward 1554 -> 1010 us; but F = 64
        327 -> 355 and F = 32 202 -> 231 us: rows of more than 64 columns only."""
        from .. import xcdplan

        nnz = self.perm.numel()
        if xcdplan.MODE == "off" or nnz == 0:
            return None
        if xcdplan.MODE != "force":
            if self.uses < 2 or nnz < xcdplan.ORDERED_MIN_EDGES or k < self.XCD_MIN_COLUMNS:
                return None
            if self._skewed is None:
                if torch.cuda.is_current_stream_capturing():
                    return None
                self._skewed = xcdplan.skewed(self.rowptr)
            if not self._skewed:
                return None
        key = tensor_key(colind)
        if key != self._xcd_key or self._xcd is None:
            if torch.cuda.is_current_stream_capturing():  # (the build reads sizes back: never inside a capture)
                return None
            if self._xcd_builds >= 4 and xcdplan.MODE != "force":
                return None  # (a caller that pairs this index with ever new partner tensors: no plan build per call)
            self._xcd_builds += 1
            split = int(_lib.hip().cogdl_hip_exact_row_edges(nnz))
            plan = xcdplan.build(self.rowptr, colind, eid_base=None if self.sorted else self.perm, split=split)
            self._xcd_key, self._xcd = key, (plan, colind)  # (colind kept: pins the key's address)
        return self._xcd[0]

    def colind(self, col):
        """The source ids in sorted order as int32, memoised on the identity (address, version, layout) of `col`; the
        memo keeps `col` alive so that its address cannot be recycled for a different tensor under the same key."""
        key = tensor_key(col)
        if key != self._col_key or self._col_src is None:
            self._colind = (col if self.sorted else col.index_select(0, self.perm.long())).int()
            self._col_key, self._col_src = key, col
        return self._colind


_PLANS = collections.OrderedDict()
_MAX_PLANS = 16


def edge_plan(dst, n):
    """Memoised on the identity + version of `dst` (Graph.edge_index hands out the same tensors call after call,
    cogdl/data/data.py:305-309)."""
    key = tensor_key(dst) + (int(n),)
    plan = _PLANS.get(key)
    if plan is None:
        plan = EdgePlan(dst, int(n))
        _PLANS[key] = plan
        while len(_PLANS) > _MAX_PLANS:
            _PLANS.popitem(last=False)
    else:
        _PLANS.move_to_end(key)
    plan.uses += 1
    return plan


def clear_plans():
    _PLANS.clear()


def _gspmm(plan, colind, x, efeat, ef_scalar, weight, op, mean, k):
    """out[v] = scale_v * sum_{j in row v} weight[id] * (x[colind[j]] OP efeat[id]); colind int32 in sorted order."""
    dev = plan.rowptr.device
    nnz = plan.perm.numel()
    out = torch.empty((plan.n, k), dtype=torch.float32, device=dev)
    xp = plan.xcd(colind, k)
    if xp is not None:
        ws, ws_bytes = _lib.workspace("cogdl_hip_gspmm_xcd_workspace_bytes", dev, xp.n_parts, k)
        with _lib.on_device(dev):
            rc = _lib.hip().cogdl_hip_gspmm_xcd(xp.ref(), _lib.ptr(plan.rowptr), _lib.ptr(x), _lib.ptr(efeat), int(ef_scalar),
                                                _lib.ptr(weight), op, int(mean), _lib.ptr(out), plan.n, k, _lib.ptr(ws), ws_bytes,
                                                _lib.stream_of(out))
        _lib.check(rc, "gspmm_xcd")
        return out
    ws, ws_bytes = _lib.workspace("cogdl_hip_gspmm_workspace_bytes", dev, nnz, k)
    eid = None if plan.sorted else plan.perm
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_gspmm(_lib.ptr(plan.rowptr), _lib.ptr(colind), _lib.ptr(eid), _lib.ptr(x),
                                        _lib.ptr(efeat), int(ef_scalar), _lib.ptr(weight), op, int(mean),
                                        _lib.ptr(out), plan.n, k, nnz, _lib.ptr(ws), ws_bytes,
                                        _lib.stream_of(out))
    _lib.check(rc, "gspmm")
    return out


def _hip_ok(*tensors):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)


class TorchRouteWarning(UserWarning):
    """A GPU call of an aggregating operator ran the reference's torch expressions (gather, multiply, scatter_add_ with
    atomics) instead of the fused HIP kernel, because its arguments are outside what the kernel covers."""


_ROUTE_NOTED = set()


def _note_torch_route(op, *tensors, why):
    """Never silent: the first GPU call per (operator, reason) that takes the torch route says so.  CPU tensors are the
    reference's own path and stay quiet."""
    if not any(torch.is_tensor(t) and t.is_cuda for t in tensors) or (op, why) in _ROUTE_NOTED:
        return
    _ROUTE_NOTED.add((op, why))
    import warnings

    warnings.warn("cogdl_amd.operators.ops.%s: GPU tensors on the reference's torch route (%s); the fused HIP kernel covers "
                  "2-D float32 features with 1-D int64 edge indices" % (op, why), TorchRouteWarning, stacklevel=3)


# --------------------------------------------------------------------------------------------- scatter_add / op_aggr
class _ScatterRows(torch.autograd.Function):
    """out[v] = scale_v * sum_{e: dst[e] == v} data[e]  (op_aggr, ops.py:28-40)."""

    @staticmethod
    def forward(ctx, data, dst, num_nodes, mean):
        plan = edge_plan(dst, num_nodes)
        data = data.contiguous()
        ctx.plan, ctx.mean = plan, mean
        ctx.save_for_backward(dst)
        # message = the edge row alone; the engine still walks a column array: the permutation serves as one
        return _gspmm(plan, plan.perm, None, data, False, None, 0, mean, data.shape[1])

    @staticmethod
    def backward(ctx, grad):
        (dst,) = ctx.saved_tensors
        if ctx.mean:
            grad = grad * _deg_inv(ctx.plan).view(-1, 1)
        return grad.index_select(0, dst), None, None, None


def _deg_inv(plan):
    deg = (plan.rowptr[1:] - plan.rowptr[:-1]).float()
    inv = deg.pow(-1)
    inv[torch.isinf(inv)] = 0
    return inv


def _scatter_add_torch(data, dst, num_nodes, dim=0):
    num_edges, num_feats = data.shape
    out = torch.zeros((num_nodes, num_feats), dtype=data.dtype, device=data.device)
    if len(dst.shape) == 1:
        dst = dst.view(-1, 1)
    return out.scatter_add_(dim=dim, index=dst.expand(num_edges, num_feats), src=data)


def _rows_index(dst, data):
    """dst as the reference accepts it (1-D, or [E, 1] to be expanded) -> 1-D int64, or None if it is a general index."""
    if dst.dim() == 2 and dst.shape[1] == 1:
        dst = dst.reshape(-1)
    if dst.dim() != 1 or dst.dtype != torch.int64 or dst.numel() != data.shape[0] or not dst.is_cuda:
        return None
    return dst


def scatter_add(data, dst, num_nodes, dim=0):
    """ops.py:4-11: zeros([num_nodes, F]).scatter_add_(dim, dst expanded over the columns, data)."""
    idx = _rows_index(dst, data) if (dim == 0 and data.dim() == 2 and _hip_ok(data)) else None
    if idx is None or data.shape[0] == 0 or data.shape[1] == 0:
        if data.numel() > 0:
            _note_torch_route("scatter_add", data, dst, why="dim %d, dtype %s, data %s, dst %s %s"
                              % (dim, data.dtype, tuple(data.shape), dst.dtype, tuple(dst.shape)))
        return _scatter_add_torch(data, dst, num_nodes, dim)
    return _ScatterRows.apply(data, idx, int(num_nodes), False)


def op_src_edge(op, src, e_feat):
    """ops.py:17-25: the message of one edge, src OP e_feat with OP in add | sub | mul."""
    if op not in _BINARY:
        raise NotImplementedError
    return _BINARY[op](src, e_feat)


def op_aggr(op, msg, dst, num_nodes):
    """ops.py:28-40: "sum" | "mean" of the messages per destination (an empty destination gives 0)."""
    if op not in ("sum", "mean"):
        raise NotImplementedError
    idx = _rows_index(dst, msg) if (msg.dim() == 2 and _hip_ok(msg)) else None
    if idx is not None and msg.shape[0] > 0 and msg.shape[1] > 0:
        return _ScatterRows.apply(msg, idx, int(num_nodes), op == "mean")
    if msg.numel() > 0:
        _note_torch_route("op_aggr", msg, dst, why="dtype %s, msg %s, dst %s %s" % (msg.dtype, tuple(msg.shape), dst.dtype, tuple(dst.shape)))
    out = _scatter_add_torch(msg, dst, num_nodes)
    if op == "mean":  # counts are exact in fp32 either way: same deg^-1 as the reference's scatter_add_ of ones
        inv = torch.bincount(dst.reshape(-1), minlength=num_nodes).float().pow(-1)
        out = out * torch.where(torch.isinf(inv), torch.zeros_like(inv), inv).view(-1, 1)
    return out


# --------------------------------------------------------------------------------------------- src OP edge -> aggregate
class _SrcOpEdgeAggr(torch.autograd.Function):
    """out[v] = scale_v * sum_{e: row[e] == v} (n_feat[col[e]] OP e_feat[e]) * data[e]  (ops.py:43-52), fused."""

    @staticmethod
    def forward(ctx, n_feat, e_feat, data, row, col, op1, mean):
        nnode = n_feat.shape[0]
        plan = edge_plan(row, nnode)
        colind = plan.colind(col)
        n_feat = n_feat.contiguous()
        e_feat = e_feat.contiguous()
        ef_scalar = e_feat.shape[1] == 1 and n_feat.shape[1] != 1
        weight = None if data is None else data.contiguous()
        ctx.plan, ctx.op1, ctx.mean, ctx.has_w = plan, op1, mean, data is not None
        ctx.save_for_backward(n_feat, e_feat, weight, row, col)
        return _gspmm(plan, colind, n_feat, e_feat, ef_scalar, weight, _OPS[op1], mean, n_feat.shape[1])

    @staticmethod
    def backward(ctx, grad):
        """Fused (no [E, F] temporaries): grad_x = the same gspmm kernel over the SOURCE-sorted view of the edges (rows =
        sources, gathered operand = the upstream gradient rows, autograd's rounding order: COGDL_HIP_GSPMM_WMUL); grad of
        the edge features / edge weights = one per-edge kernel (cogdl_hip_gspmm_edge_grad).  Per element the edges are
        added in the caller's edge order: what index_add_ does on the CPU (the reference's gradients, bit for bit, while
        a source has at most `long-row threshold` out-edges); on a GPU the reference's scatter has no order at all."""
        n_feat, e_feat, weight, row, col = ctx.saved_tensors
        grad = grad.contiguous()
        scale = _deg_inv(ctx.plan) if ctx.mean else None
        op = _OPS[ctx.op1]
        ef_scalar = e_feat.shape[1] == 1 and n_feat.shape[1] != 1
        k = n_feat.shape[1]
        g_x = g_e = g_w = None
        if ctx.needs_input_grad[0]:
            gs = grad * scale.view(-1, 1) if scale is not None else grad  # [n, k]: autograd's `grad * deg_inv`
            splan = edge_plan(col, n_feat.shape[0])  # source-sorted view (memoised like the destination plan)
            dst_sorted = splan.colind(row)
            if ctx.op1 == "mul":
                g_x = _gspmm(splan, dst_sorted, gs, e_feat, ef_scalar, weight, _OPS_WMUL, False, k)
            else:  # d msg / d src = 1: the message is the gradient row alone (times the weight)
                g_x = _gspmm(splan, dst_sorted, gs, None, False, weight, 0, False, k)
        need_e, need_w = ctx.needs_input_grad[1], ctx.has_w and ctx.needs_input_grad[2]
        if need_e or need_w:
            dev = grad.device
            e = row.numel()
            if need_e:
                g_e = torch.empty((e, 1) if ef_scalar else (e, k), dtype=torch.float32, device=dev)
            if need_w:
                g_w = torch.empty(e, dtype=torch.float32, device=dev)
            # (bound to locals: a temporary copy of a strided row / col would go back to the caching allocator as soon as its
            #  address is taken, and the second copy would most likely reuse the first one's block before the launch)
            row_c, col_c = row.contiguous(), col.contiguous()
            with _lib.on_device(dev):
                rc = _lib.hip().cogdl_hip_gspmm_edge_grad(_lib.ptr(row_c), _lib.ptr(col_c), _lib.ptr(grad), _lib.ptr(scale),
                                                          _lib.ptr(weight), _lib.ptr(n_feat), _lib.ptr(e_feat), int(ef_scalar),
                                                          op, _lib.ptr(g_e), _lib.ptr(g_w), e, k, _lib.stream_of(grad))
            _lib.check(rc, "gspmm_edge_grad")
            if need_e and e_feat.shape[1] == 1 and not ef_scalar:  # k == 1: [E, 1] either way
                g_e = g_e.view(e, 1)
        return g_x, g_e, g_w, None, None, None, None


def src_op_e_aggr_coo(op1, op2, n_feat, e_feat, row, col, data=None):
    """ops.py:43-52: out = aggr_{op2 in sum|mean} over edges (row <- col) of (n_feat[col] op1 e_feat) * data."""
    nnode = n_feat.shape[0]
    if len(e_feat.shape) == 1:
        e_feat = e_feat.view(-1, 1)
    fused = (op1 in _OPS and op2 in ("sum", "mean") and n_feat.dim() == 2 and e_feat.dim() == 2
             and _hip_ok(n_feat, e_feat, data) and row.is_cuda and row.dtype == torch.int64 and row.dim() == 1
             and col.dtype == torch.int64 and e_feat.shape[0] == row.numel() and row.numel() > 0
             and n_feat.shape[1] > 0 and e_feat.shape[1] in (1, n_feat.shape[1])
             and (data is None or (data.dim() == 1 and data.numel() == row.numel())))
    if fused:
        return _SrcOpEdgeAggr.apply(n_feat, e_feat, data, row, col, op1, op2 == "mean")
    if row.numel() > 0:
        _note_torch_route("s_%s_e_%s" % (op1, op2), n_feat, e_feat, row,
                          why="n_feat %s %s, e_feat %s %s, row %s, col %s, data %s"
                          % (n_feat.dtype, tuple(n_feat.shape), e_feat.dtype, tuple(e_feat.shape), row.dtype, col.dtype,
                             None if data is None else tuple(data.shape)))
    src = n_feat[col]
    msg = op_src_edge(op1, src, e_feat)
    if data is not None:
        msg = msg * data.view(-1, 1)
    return op_aggr(op2, msg, row, nnode)


def _named(fn, name, doc):
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = doc
    return fn


def _make_aggr(op1, op2, lines):
    def f(g, n_feat, e_feat, weight=False):
        row, col = g.edge_index
        return src_op_e_aggr_coo(op1, op2, n_feat, e_feat, row, col, data=g.edge_weight if weight else None)

    return _named(f, "s_%s_e_%s" % (op1, op2),
                  "out[v] = %s over the edges (v <- u) of (n_feat[u] %s e_feat[e]) [* g.edge_weight[e]]  (ops.py:%s)."
                  % (op2, op1, lines))


s_add_e_sum = _make_aggr("add", "sum", "55-61")
s_mul_e_sum = _make_aggr("mul", "sum", "64-70")
s_sub_e_sum = _make_aggr("sub", "sum", "73-79")
s_add_e_mean = _make_aggr("add", "mean", "82-88")
s_mul_e_mean = _make_aggr("mul", "mean", "91-97")
s_sub_e_mean = _make_aggr("sub", "mean", "100-106")


# --------------------------------------------------------------------------------------------- elementwise operators
# One gather and one arithmetic op per element: plain torch, exactly the reference's expressions (bit-identical).
_BINARY = {"add": torch.add, "sub": torch.sub, "mul": torch.mul}


def _make_src_edge(op, lines):
    def f(g, n_feat, e_feat):
        return _BINARY[op](n_feat[g.edge_index[1]], e_feat)

    return _named(f, "s_%s_e" % op, "[E, F] message n_feat[col] %s e_feat  (ops.py:%s)." % (op, lines))


s_add_e = _make_src_edge("add", "112-114")
s_sub_e = _make_src_edge("sub", "117-119")
s_mul_e = _make_src_edge("mul", "122-124")


def src_op_target_coo(op, g, src, tgt):
    """ops.py:130-147: source row OP target row per edge; tgt None: both ends are gathered from `src`."""
    if tgt is None:
        row, col = g.edge_index
        src, tgt = src[col], src[row]
    if op in _BINARY:
        return _BINARY[op](src, tgt)
    if op == "dot":
        return (src * tgt).sum(1, keepdim=True)
    if op == "div":
        out = src / tgt
        out[torch.isinf(out)] = 0  # x / 0 -> 0 like the reference (0 / 0 stays nan, ops.py:143-145)
        return out
    raise NotImplementedError


def _make_src_target(op, lines):
    def f(g, src, dst=None):
        return src_op_target_coo(op, g, src, dst)

    return _named(f, "s_%s_t" % op, "per edge: src[col] %s (dst if given else src[row])  (ops.py:%s)." % (op, lines))


s_add_t = _make_src_target("add", "150-151")
s_mul_t = _make_src_target("mul", "154-155")
s_sub_t = _make_src_target("sub", "158-159")
s_dot_t = _make_src_target("dot", "162-163")
s_div_t = _make_src_target("div", "166-167")


def message_passing(send_func, msg_func, rec_fun):
    """ops.py:170-171: a stub in the reference as well."""
