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
