"""Drop-in for cogdl/operators/sample.py: `sample_adj_c`, `subgraph_c`, `coo2csr_cpu`,
`coo2csr_cpu_index` (operators/sample.py:8-12) on libcogdl_host.so -- HIP-free, fork-safe.
`sample_adj_c` additionally accepts a graph that lives on the GPU (indptr/indices CUDA tensors): sampling and
relabelling then run there (cogdl_hip_sample_adj, csrc/sample.hip) and the results stay on the GPU; libcogdl_hip.so
is only loaded on that path, so CPU callers -- CogDL's forked DataLoader workers -- never touch the HIP runtime.

Semantics follow cogdl/operators/sample/sample.cpp; differences:
  * non-contiguous inputs are made contiguous (the reference reads raw data_ptr and silently
    returns a wrong CSR for `edge_index.t()` views, sample.cpp:242-243);
  * out-of-range node ids raise instead of corrupting memory;
  * random modes draw from an explicit, reproducible generator: the seed is taken from torch's
    CPU generator (so `torch.manual_seed` controls sampling) instead of unseeded libc rand().
"""
import os
import threading

import torch
import torch.utils.data

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


def _sample_adj_gpu(indptr, indices, node_idx, num_neighbors, replace, seed):
    """Graph on the GPU -> (row_ptr, col, nodes, edges) on the GPU.  row_ptr comes back already padded to
    len(nodes) + 1 entries: Graph.sample_adj (cogdl/data/data.py:828-830) would otherwise build that padding on the
    CPU and fail to concatenate it with a GPU tensor."""
    dev = indptr.device
    indptr, indices = indptr.to(torch.long).contiguous(), indices.to(torch.long).contiguous()
    if not torch.is_tensor(node_idx):
        node_idx = torch.as_tensor(node_idx, dtype=torch.long)
    node_idx = node_idx.to(device=dev, dtype=torch.long).contiguous()
    n, b = indptr.numel() - 1, node_idx.numel()
    num_neighbors = int(num_neighbors)
    if num_neighbors < 0:  # capacity = the seeds' total degree (ids clamped here: the kernel reports bad ones)
        safe = node_idx.clamp(0, max(n - 1, 0))
        cap_e = int((indptr[safe + 1] - indptr[safe]).sum()) if b and n else 0
    else:
        cap_e = b * num_neighbors
    if seed is None:
        seed = 0 if num_neighbors < 0 else int(torch.randint(0, 2 ** 62, (1,)).item())
    out_indptr = torch.empty(b + 1, dtype=torch.long, device=dev)
    out_indices = torch.empty(cap_e, dtype=torch.long, device=dev)
    out_nodes = torch.empty(b + cap_e, dtype=torch.long, device=dev)
    out_edges = torch.empty(cap_e, dtype=torch.long, device=dev)
    counts = torch.empty(3, dtype=torch.long, device=dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_sample_adj_workspace_bytes(b, cap_e, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.cogdl_hip_sample_adj(_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b, num_neighbors,
                                      int(bool(replace)), seed, _lib.ptr(out_indptr), _lib.ptr(out_indices),
                                      _lib.ptr(out_nodes), _lib.ptr(out_edges), cap_e, _lib.ptr(counts),
                                      _lib.ptr(ws), ws_bytes, _lib.stream_of(indptr))
    _lib.check(rc, "sample_adj")
    nn, ne, flags = (int(v) for v in counts.tolist())  # the one synchronisation: output sizes
    if flags & 1:
        raise _lib.BackendError("sample_adj: seed node id out of range [0, %d)" % n)
    if flags:
        raise _lib.BackendError("sample_adj: %s" % ("neighbour id out of range" if flags & 2 else "capacity exceeded"))
    if nn > b:  # pad like data.py:828-830 does: every non-seed node is a row without edges
        out_indptr = torch.cat([out_indptr, out_indptr[-1:].expand(nn - b)])
    return out_indptr, out_indices[:ne], out_nodes[:nn], out_edges[:ne]


def _sampler_threads():
    """OpenMP threads for the pick phase of the host sampler: COGDL_AMD_SAMPLER_THREADS, default 1.  Opt-in because the
    gain depends on the host: 8 threads made two-hop batches 1.85x (1024 seeds) / 3x (8192) faster on an 8-core box, but
    on the 256-core NUMA host of the MI355X box they were 3.5x SLOWER at 1024 seeds (1.7 -> 5.9 ms; 8192 seeds:
    noisy either way) -- one thread there already is 30x the reference's sampler.  DataLoader workers are the
    parallelism of the reference's own pipeline in any case."""
    env = os.environ.get("COGDL_AMD_SAMPLER_THREADS")
    if not env:
        return 1
    # A forked DataLoader worker must never enter an OpenMP region: its parent may have used libgomp already, whose
    # thread pool does not survive fork() (the classic hang) -- workers are the pipeline's parallelism anyway.
    if torch.utils.data.get_worker_info() is not None:
        return 1
    return max(1, min(int(env), torch.get_num_threads()))


def sample_adj_padded(indptr, indices, node_idx, num_neighbors, replace=False, seed=0, seed_dev=None, count=None,
                      counts_out=None, block32=False):
    """sample_adj into buffers of FIXED capacity (GPU graphs only): no size depends on what was sampled and nothing
    synchronises, so the call can sit inside a captured hipGraph (cogdl_amd.graphs.capture / torch.cuda.graph).

        node_idx : [B] int64 seed slots, of which `count` (a device int64 scalar tensor; None = all B) are in use
        seed_dev : device int64 scalar tensor added to `seed` on the device at every launch (a captured step bumps it
                   in place between replays); None = `seed` alone
        counts_out : a [3] int64 tensor on the GPU to receive the counts (a row of a caller's table); None = a new one
        block32  : also produce the block as the SpMM takes it -- (rowptr int32 [B + 1], col int32 [B*k], 1 / in-degree
                   float32 [B]) -- in the sampler's own launches (cogdl_hip_sample_adj_block); it is attached to the
                   returned row_ptr, where graph_build.block_for_spmm finds it instead of launching its conversion
    Returns (row_ptr [B + B*k + 1], col [B*k], nodes [B + B*k], edges [B*k], counts [3] = {N', E', flags}) -- all on the
    GPU, the unused tails filled as cogdl_hip_sample_adj_padded documents (empty rows, index 0).  The valid prefix is
    exactly what sample_adj_c returns for the same seed."""
    dev = indptr.device
    if not (torch.is_tensor(indptr) and indptr.is_cuda):
        raise _lib.BackendError("sample_adj_padded: the graph must live on the GPU")
    num_neighbors = int(num_neighbors)
    if num_neighbors < 0:
        raise _lib.BackendError("sample_adj_padded: a fixed capacity needs num_neighbors >= 0")
    for name, t in (("indptr", indptr), ("indices", indices), ("node_idx", node_idx)):
        if t.dtype != torch.long or not t.is_contiguous() or t.device != dev:
            raise _lib.BackendError("sample_adj_padded: %s must be a contiguous int64 tensor on %s" % (name, dev))
    for name, t in (("count", count), ("seed_dev", seed_dev)):  # read as 8-byte words on the device
        if t is not None and (not torch.is_tensor(t) or t.dtype != torch.long or t.device != dev or t.numel() < 1
                              or not t.is_contiguous()):
            raise _lib.BackendError("sample_adj_padded: %s must be an int64 tensor on %s (a device scalar)" % (name, dev))
    n, b = indptr.numel() - 1, node_idx.numel()
    cap_e = b * num_neighbors
    out_indptr = torch.empty(b + cap_e + 1, dtype=torch.long, device=dev)
    out_indices = torch.empty(cap_e, dtype=torch.long, device=dev)
    out_nodes = torch.empty(b + cap_e, dtype=torch.long, device=dev)
    out_edges = torch.empty(cap_e, dtype=torch.long, device=dev)
    if counts_out is None:
        counts = torch.empty(3, dtype=torch.long, device=dev)
    else:
        counts = counts_out
        if (not torch.is_tensor(counts) or counts.dtype != torch.long or counts.device != dev or counts.numel() != 3
                or not counts.is_contiguous()):
            raise _lib.BackendError("sample_adj_padded: counts_out must be a contiguous int64 tensor of 3 elements on %s" % dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_sample_adj_workspace_bytes(b, cap_e, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    head = (_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b, _lib.ptr(count), num_neighbors,
            int(bool(replace)), int(seed), _lib.ptr(seed_dev), _lib.ptr(out_indptr), _lib.ptr(out_indices),
            _lib.ptr(out_nodes), _lib.ptr(out_edges), cap_e, _lib.ptr(counts))
    with _lib.on_device(dev):
        if block32:
            rp32 = torch.empty(b + 1, dtype=torch.int32, device=dev)
            col32 = torch.empty(cap_e, dtype=torch.int32, device=dev)
            inv = torch.empty(b, dtype=torch.float32, device=dev)
            rc = lib.cogdl_hip_sample_adj_block(*head, _lib.ptr(rp32), _lib.ptr(col32), _lib.ptr(inv), _lib.ptr(ws), ws_bytes,
                                                _lib.stream_of(indptr))
            rp32._cogdl_max_row_edges = num_neighbors  # (csrspmm_block: no row reaches the long-row threshold)
            out_indptr._cogdl_block32 = (out_indices, rp32, col32, inv)
        else:
            rc = lib.cogdl_hip_sample_adj_padded(*head, _lib.ptr(ws), ws_bytes, _lib.stream_of(indptr))
    _lib.check(rc, "sample_adj_padded")
    return out_indptr, out_indices, out_nodes, out_edges, counts


_SCRATCH = threading.local()


def _host_scratch(cap_e, cap_n):
    """(out_indices[cap_e], out_nodes[cap_n], out_edges[cap_e]) views of this thread's reusable int64 scratch; the
    caller copies what was actually produced out of it before the next call."""
    need = 2 * cap_e + cap_n
    buf = getattr(_SCRATCH, "buf", None)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 16), dtype=torch.long)
        _SCRATCH.buf = buf
    return buf[:cap_e], buf[cap_e:cap_e + cap_n], buf[cap_e + cap_n:cap_e + cap_n + cap_e]


def sample_adj_c(indptr, indices, node_idx, num_neighbors, replace, seed=None):
    if torch.is_tensor(indptr) and indptr.is_cuda:
        return _sample_adj_gpu(indptr, indices, node_idx, num_neighbors, replace, seed)
    indptr, indices, node_idx = _i64(indptr), _i64(indices), _i64(node_idx)
    n, b = indptr.numel() - 1, node_idx.numel()
    num_neighbors = int(num_neighbors)
    if b and (int(node_idx.min()) < 0 or int(node_idx.max()) >= n):
        raise _lib.BackendError("sample_adj: seed node id out of range [0, %d)" % n)
    if num_neighbors < 0:
        cap_e = int((indptr[node_idx + 1] - indptr[node_idx]).sum())
    else:  # an upper bound is enough (outputs are trimmed to the counts the library reports): no gathers of indptr here
        cap_e = b * num_neighbors
    cap_n = b + cap_e
    if seed is None:
        seed = 0 if num_neighbors < 0 else int(torch.randint(0, 2 ** 62, (1,)).item())
    out_indptr = torch.empty(b + 1, dtype=torch.long)
    # capacity-sized scratch, kept per thread and grown on demand: a second hop asks for tens of MB of upper-bound
    # capacity of which a fraction is used -- fresh allocations of that size are mmap'ed and page-faulted on every call
    out_indices, out_nodes, out_edges = _host_scratch(cap_e, cap_n)
    counts = torch.zeros(2, dtype=torch.long)
    rc = _lib.host().cogdl_host_sample_adj_mt(_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b,
                                              num_neighbors, int(bool(replace)), seed, _lib.ptr(out_indptr),
                                              _lib.ptr(out_indices), _lib.ptr(out_nodes), _lib.ptr(out_edges), cap_e,
                                              cap_n, _lib.ptr(counts), _sampler_threads())
    _lib.check_host(rc, "sample_adj")
    nn, ne = int(counts[0]), int(counts[1])
    return out_indptr, out_indices[:ne].clone(), out_nodes[:nn].clone(), out_edges[:ne].clone()


def _subgraph_gpu(indptr, indices, node_idx):
    """Graph on the GPU -> (row_ptr, col, nodes, edges) on the GPU (cogdl_hip_subgraph, csrc/subgraph.hip).  node_idx
    may arrive on the CPU: Graph.csr_subgraph moves it there (cogdl/data/data.py:853-854) before calling."""
    dev = indptr.device
    indptr, indices = indptr.to(torch.long).contiguous(), indices.to(torch.long).contiguous()
    if not torch.is_tensor(node_idx):
        node_idx = torch.as_tensor(node_idx, dtype=torch.long)
    node_idx = node_idx.to(device=dev, dtype=torch.long).contiguous()
    n, b = indptr.numel() - 1, node_idx.numel()
    safe = node_idx.clamp(0, max(n - 1, 0))
    cap_e = int((indptr[safe + 1] - indptr[safe]).sum()) if b and n else 0  # upper bound (ids clamped: the kernel flags bad ones)
    out_indptr = torch.empty(b + 1, dtype=torch.long, device=dev)
    out_indices = torch.empty(cap_e, dtype=torch.long, device=dev)
    out_edges = torch.empty(cap_e, dtype=torch.long, device=dev)
    counts = torch.empty(2, dtype=torch.long, device=dev)
    lib = _lib.hip()
    ws_bytes = lib.cogdl_hip_subgraph_workspace_bytes(b, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        rc = lib.cogdl_hip_subgraph(_lib.ptr(indptr), _lib.ptr(indices), n, _lib.ptr(node_idx), b, _lib.ptr(out_indptr),
                                    _lib.ptr(out_indices), _lib.ptr(out_edges), cap_e, _lib.ptr(counts), _lib.ptr(ws),
                                    ws_bytes, _lib.stream_of(indptr))
    _lib.check(rc, "subgraph")
    ne, flags = (int(v) for v in counts.tolist())  # the one synchronisation: the output size
    if flags & 1:
        raise _lib.BackendError("subgraph: node id out of range [0, %d)" % n)
    if flags:
        raise _lib.BackendError("subgraph: %s" % ("neighbour id out of range" if flags & 2 else "capacity exceeded"))
    return out_indptr, out_indices[:ne], torch.arange(0, b, device=dev), out_edges[:ne]


def subgraph_c(indptr, indices, node_idx):
    if torch.is_tensor(indptr) and indptr.is_cuda:
        return _subgraph_gpu(indptr, indices, node_idx)
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
