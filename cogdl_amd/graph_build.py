"""GPU-resident CSR construction: `coo2csr_index` with the signature and results of
cogdl/utils/graph_utils.py:133-142, but without its GPU -> CPU -> GPU round trip: tensors on the GPU are sorted
there (cogdl_hip_coo2csr_index: radix sort of the row keys), CPU tensors go to the host operator
(cogdl_host_coo2csr_index, the counterpart of sample.cpp:234-270).  `install()` rebinds CogDL's function to this one.
"""
import torch

from . import _lib


def coo2csr_index(row, col, num_nodes=None):
    """-> (row_ptr int64 [num_nodes+1], reindex int64 [nnz]) on row's device; edges of a row keep their COO order."""
    if num_nodes is None:
        num_nodes = int(torch.max(torch.stack([row, col])).item()) + 1 if row.numel() else 0
    num_nodes = int(num_nodes)
    if not row.is_cuda:
        from .operators.sample import coo2csr_cpu_index

        return coo2csr_cpu_index(row, col, num_nodes)
    dev = row.device
    row = row.long().contiguous()
    nnz = row.numel()
    row_ptr = torch.empty(num_nodes + 1, dtype=torch.long, device=dev)
    perm = torch.empty(nnz, dtype=torch.long, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_coo2csr_index_workspace_bytes(nnz, num_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.cogdl_hip_coo2csr_index(_lib.ptr(row), nnz, num_nodes, _lib.ptr(row_ptr), _lib.ptr(perm), _lib.ptr(bad),
                                         _lib.ptr(ws), ws_bytes, _lib.stream_of(row))
    _lib.check(rc, "coo2csr_index")
    if int(bad.item()):  # the reference would corrupt memory here
        raise _lib.BackendError("coo2csr_index: a row id lies outside [0, %d)" % num_nodes)
    return row_ptr, perm


# ----------------------------------------------------------------------------------------------------------------------
# add_remaining_self_loops / symmetric_normalization / row_normalization with the reference's signatures
# (cogdl/utils/graph_utils.py:40-89): GPU tensors go to the HIP kernels of csrc/graph_norm.hip, everything else (CPU
# tensors, non-float weights) to the reference's own torch expressions, restated below.
def _gpu_coo(row, col, val):
    """Do the HIP kernels take this call?  They sit outside autograd, while the reference's torch expressions are
    differentiable in the weights: a weight tensor that takes part in autograd (learned edge weights) keeps the
    reference's path and with it its graph."""
    return (torch.is_tensor(row) and row.is_cuda and row.dtype == torch.long and col.dtype == torch.long and row.dim() == 1
            and (val is None or (val.is_cuda and val.dtype == torch.float32 and val.dim() == 1
                                 and val.numel() == row.numel()
                                 and not (val.requires_grad and torch.is_grad_enabled()))))


def _check_bad(bad, what, n):
    if int(bad.item()):
        raise _lib.BackendError("%s: an index lies outside [0, %d)" % (what, n))


def add_remaining_self_loops(edge_index, edge_weight=None, fill_value=1, num_nodes=None):
    """graph_utils.py:40-70 -> ((row, col), weight): existing self loops dropped, one loop per node appended; a node
    that had a loop keeps (the last of) its loop weights, the others get fill_value."""
    row, col = edge_index[0], edge_index[1]
    if fill_value is None:
        fill_value = 1
    if num_nodes is None:
        num_nodes = max(row.max().item(), col.max().item()) + 1
    num_nodes = int(num_nodes)
    if not _gpu_coo(row, col, edge_weight):
        if edge_weight is None:
            edge_weight = torch.ones(row.shape[0], device=row.device)
        mask = row != col
        loops = torch.arange(0, num_nodes, dtype=row.dtype, device=row.device)
        loop_weight = torch.full((num_nodes,), fill_value, dtype=edge_weight.dtype, device=edge_weight.device)
        rest = edge_weight[~mask]
        if rest.numel() > 0:
            loop_weight[row[~mask]] = rest
        return (torch.cat([row[mask], loops]), torch.cat([col[mask], loops])), torch.cat([edge_weight[mask], loop_weight])
    dev = row.device
    row, col = row.contiguous(), col.contiguous()
    val = None if edge_weight is None else edge_weight.contiguous()
    nnz = row.numel()
    out_row = torch.empty(nnz + num_nodes, dtype=torch.long, device=dev)
    out_col = torch.empty(nnz + num_nodes, dtype=torch.long, device=dev)
    out_val = torch.empty(nnz + num_nodes, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.long, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_add_remaining_self_loops_workspace_bytes(nnz, num_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.cogdl_hip_add_remaining_self_loops(_lib.ptr(row), _lib.ptr(col), _lib.ptr(val), nnz, num_nodes,
                                                    float(fill_value), _lib.ptr(out_row), _lib.ptr(out_col),
                                                    _lib.ptr(out_val), _lib.ptr(count), _lib.ptr(bad), _lib.ptr(ws),
                                                    ws_bytes, _lib.stream_of(row))
    _lib.check(rc, "add_remaining_self_loops")
    n_out = int(count.item())  # the one synchronisation: the output size
    _check_bad(bad, "add_remaining_self_loops", num_nodes)
    return (out_row[:n_out], out_col[:n_out]), out_val[:n_out]


def _norm_weights(num_nodes, row, col, val, mode, what):
    dev = row.device
    row, col = row.contiguous(), col.contiguous()
    val = None if val is None else val.contiguous()
    nnz, num_nodes = row.numel(), int(num_nodes)
    out = torch.empty(nnz, dtype=torch.float32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_coo_norm_weights_workspace_bytes(num_nodes)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.cogdl_hip_coo_norm_weights(_lib.ptr(row), _lib.ptr(col), _lib.ptr(val), nnz, num_nodes, mode,
                                            _lib.ptr(out), _lib.ptr(bad), _lib.ptr(ws), ws_bytes, _lib.stream_of(row))
    _lib.check(rc, what)
    _check_bad(bad, what, num_nodes)
    return out


def symmetric_normalization(num_nodes, row, col, val=None):
    """graph_utils.py:82-89: d^-1/2[col] * val * d^-1/2[row], d = number of edges per row."""
    if not _gpu_coo(row, col, val):
        if val is None:
            val = torch.ones(row.shape[0]).to(row.device)
        d = torch.zeros(num_nodes, device=row.device).scatter_add_(0, row, torch.ones(col.shape[0], device=row.device))
        dinv = d.pow(-0.5)
        dinv[dinv == float("inf")] = 0
        return dinv[col] * val * dinv[row]
    return _norm_weights(num_nodes, row, col, val, 0, "symmetric_normalization")


def row_normalization(num_nodes, row, col, val=None):
    """graph_utils.py:72-79: val / d[row]."""
    if not _gpu_coo(row, col, val):
        if val is None:
            val = torch.ones(row.shape[0], device=row.device)
        d = torch.zeros(num_nodes, device=row.device).scatter_add_(0, row, torch.ones(col.shape[0], device=row.device))
        dinv = d.pow(-1).view(-1)
        dinv[torch.isinf(dinv)] = 0
        return val * dinv[row]
    return _norm_weights(num_nodes, row, col, val, 1, "row_normalization")


def block_for_spmm(row_ptr, col, n_rows=None, mean=True):
    """A sampled block (int64 row_ptr / col on the GPU, as sample_adj_c / sample_adj_padded return it) -> (rowptr int32
    [n_rows + 1], col int32, in_norm float32 [n_rows] | None) in ONE launch (cogdl_hip_block_prepare): the two .int()
    copies of the dispatcher (cogdl/utils/spmm_utils.py:106) and, with `mean`, the 1 / in-degree vector of
    Graph.row_norm() (cogdl/data/data.py:240-258; 0 for a row without edges).  n_rows < the block's rows keeps only the
    target rows (graphsage.py:99 drops the others after the layer anyway)."""
    dev = _lib.require_cuda(row_ptr, col)
    if row_ptr.dtype != torch.long or col.dtype != torch.long:
        raise _lib.BackendError("block_for_spmm expects the sampler's int64 row_ptr / col (got %s / %s)"
                                % (row_ptr.dtype, col.dtype))
    row_ptr, col = row_ptr.contiguous(), col.contiguous()
    m = row_ptr.numel() - 1 if n_rows is None else int(n_rows)
    if m < 0 or m > row_ptr.numel() - 1:
        raise _lib.BackendError("block_for_spmm: n_rows = %d outside the block's %d rows" % (m, row_ptr.numel() - 1))
    ready = getattr(row_ptr, "_cogdl_block32", None)  # sample_adj_padded(block32=True): the sampler wrote it already
    if ready is not None and ready[0] is col and ready[1].numel() == m + 1:
        return ready[1], ready[2], (ready[3] if mean else None)
    rp32 = torch.empty(m + 1, dtype=torch.int32, device=dev)
    col32 = torch.empty(col.numel(), dtype=torch.int32, device=dev)
    inv = torch.empty(m, dtype=torch.float32, device=dev) if mean else None
    with _lib.on_device(dev):
        rc = _lib.hip().cogdl_hip_block_prepare(_lib.ptr(row_ptr), _lib.ptr(col), m, col.numel(), _lib.ptr(rp32),
                                                _lib.ptr(col32), _lib.ptr(inv), _lib.stream_of(row_ptr))
    _lib.check(rc, "block_prepare")
    return rp32, col32, inv
