"""Drop-in for cogdl/operators/sample.py: `sample_adj_c`, `subgraph_c`, `coo2csr_cpu`,
`coo2csr_cpu_index` (operators/sample.py:8-12) on libcogdl_host.so -- HIP-free, fork-safe.

Semantics follow cogdl/operators/sample/sample.cpp; differences:
  * non-contiguous inputs are made contiguous (the reference reads raw data_ptr and silently
    returns a wrong CSR for `edge_index.t()` views, sample.cpp:242-243);
  * out-of-range node ids raise instead of corrupting memory;
  * random modes draw from an explicit, reproducible generator: the seed is taken from torch's
    CPU generator (so `torch.manual_seed` controls sampling) instead of unseeded libc rand().
"""
import torch

from .. import _lib

_lib.host()


def _i64(t):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t, dtype=torch.long)
    if t.device.type != "cpu":
        raise _lib.BackendError("sampler operators take CPU tensors (got %s)" % t.device)
    return t.to(torch.long).contiguous()


def coo2csr_cpu(row, col, val, num_nodes):
    row, col = _i64(row), _i64(col)
    val = val.to(torch.float32).contiguous()
    nnz = row.numel()
    row_ptr = torch.empty(num_nodes + 1, dtype=torch.long)
    col_ind = torch.empty(nnz, dtype=torch.long)
    out_val = torch.empty(nnz, dtype=torch.float32)
    rc = _lib.host().cogdl_host_coo2csr(_lib.ptr(row), _lib.ptr(col), _lib.ptr(val), nnz, int(num_nodes),
                                        _lib.ptr(row_ptr), _lib.ptr(col_ind), _lib.ptr(out_val))
    _lib.check_host(rc, "coo2csr_cpu")
    return row_ptr, col_ind, out_val


def coo2csr_cpu_index(row, col, num_nodes):
    row = _i64(row)
    nnz = row.numel()
    row_ptr = torch.empty(num_nodes + 1, dtype=torch.long)
    perm = torch.empty(nnz, dtype=torch.long)
    rc = _lib.host().cogdl_host_coo2csr_index(_lib.ptr(row), nnz, int(num_nodes), _lib.ptr(row_ptr), _lib.ptr(perm))
    _lib.check_host(rc, "coo2csr_cpu_index")
    return row_ptr, perm


def sample_adj_c(indptr, indices, node_idx, num_neighbors, replace, seed=None):
    indptr, indices, node_idx = _i64(indptr), _i64(indices), _i64(node_idx)
    n, b = indptr.numel() - 1, node_idx.numel()
    num_neighbors = int(num_neighbors)
    if b and (int(node_idx.min()) < 0 or int(node_idx.max()) >= n):
        raise _lib.BackendError("sample_adj: seed node id out of range [0, %d)" % n)
    deg = indptr[node_idx + 1] - indptr[node_idx]
    if num_neighbors < 0:
        cap_e = int(deg.sum())
    elif replace:
        cap_e = int((deg > 0).sum()) * num_neighbors
    else:
        cap_e = int(torch.clamp(deg, max=num_neighbors).sum())
    cap_n = b + cap_e
    if seed is None:
        seed = 0 if num_neighbors < 0 else int(torch.randint(0, 2 ** 62, (1,)).item())
    out_indptr = torch.empty(b + 1, dtype=torch.long)
    out_indices = torch.empty(cap_e, dtype=torch.long)
    out_nodes = torch.empty(cap_n, dtype=torch.long)
    out_edges = torch.empty(cap_e, dtype=torch.long)
    counts = torch.zeros(2, dtype=torch.long)
    rc = _lib.host().cogdl_host_sample_adj(_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b,
                                           num_neighbors, int(bool(replace)), seed, _lib.ptr(out_indptr),
                                           _lib.ptr(out_indices), _lib.ptr(out_nodes), _lib.ptr(out_edges), cap_e,
                                           cap_n, _lib.ptr(counts))
    _lib.check_host(rc, "sample_adj")
    nn, ne = int(counts[0]), int(counts[1])
    return out_indptr, out_indices[:ne].clone(), out_nodes[:nn].clone(), out_edges[:ne].clone()


def subgraph_c(indptr, indices, node_idx):
    indptr, indices, node_idx = _i64(indptr), _i64(indices), _i64(node_idx)
    n, b = indptr.numel() - 1, node_idx.numel()
    cap_e = int((indptr[node_idx + 1] - indptr[node_idx]).sum()) if b else 0
    out_indptr = torch.empty(b + 1, dtype=torch.long)
    out_indices = torch.empty(cap_e, dtype=torch.long)
    out_edges = torch.empty(cap_e, dtype=torch.long)
    counts = torch.zeros(1, dtype=torch.long)
    rc = _lib.host().cogdl_host_subgraph(_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b,
                                         _lib.ptr(out_indptr), _lib.ptr(out_indices), _lib.ptr(out_edges), cap_e,
                                         _lib.ptr(counts))
    _lib.check_host(rc, "subgraph")
    ne = int(counts[0])
    return out_indptr, out_indices[:ne].clone(), torch.arange(0, b), out_edges[:ne].clone()
