"""ctypes binding of the two C-ABI libraries (include/cogdl_hip.h, include/cogdl_host.h).

Loading is strict: if libcogdl_hip.so is missing or fails to load, importing any GPU operator
raises -- there is no CPU or PyTorch fallback behind the HIP entry points (the reference, by
contrast, swallows build failures and silently drops to torch.scatter_add:
cogdl/operators/spmm.py:30-31).
"""
import ctypes
import os

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HIP_LIB_PATH = os.path.join(_CSRC, "libcogdl_hip.so")
HOST_LIB_PATH = os.path.join(_CSRC, "libcogdl_host.so")

_vp, _i64, _i32, _f32, _sz, _u64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                     ctypes.c_size_t, ctypes.c_uint64)

# name -> (argtypes, restype): must list every symbol declared in include/cogdl_hip.h
HIP_SIGNATURES = {
    "cogdl_hip_abi_version": ([], _i32),
    "cogdl_hip_strerror": ([_i32], ctypes.c_char_p),
    "cogdl_hip_last_hip_error": ([], _i32),
    "cogdl_hip_set_tuning": ([_i32, _i32], _i32),
    "cogdl_hip_probe_read_stream": ([_vp, _sz, _vp, _vp], _i32),
    "cogdl_hip_probe_copy_stream": ([_vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_workspace_bytes": ([_i64, _i64, _i32], _sz),
    "cogdl_hip_long_row_threshold": ([_i64], _i32),
    "cogdl_hip_exact_row_edges": ([_i64], _i32),
    "cogdl_hip_csr_spmm": ([_vp] * 5 + [_i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_acc": ([_vp] * 5 + [_i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_xcd_workspace_bytes": ([_i64, _i64, _i32], _sz),
    "cogdl_hip_csr_spmm_xcd": ([_vp] * 4 + [_i64, _i64, _i32, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_gat_fwd_xcd_workspace_bytes": ([_i64, _i64, _i64, _i32], _sz),
    "cogdl_hip_gat_fwd_xcd": ([_vp] * 4 + [_f32, _f32, _u64] + [_vp] * 3 + [_i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_gat_bwd_xcd_workspace_bytes": ([_i64, _i64, _i64, _i64, _i64, _i32], _sz),
    "cogdl_hip_gat_bwd_xcd": ([_vp] * 5 + [_f32, _f32, _u64] + [_vp] * 8 + [_sz, _i64, _i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_csr_spmm_variant": ([_vp] * 5 + [_i64, _i64, _i64, _i32, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_epilogue": ([_vp] * 5 + [_i64, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr2csc_workspace_bytes": ([_i64, _i64, _i64], _sz),
    "cogdl_hip_csr2csc": ([_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr2csc_padded": ([_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr2csc_padded_workspace_bytes": ([_i64, _i64, _i64], _sz),
    "cogdl_hip_gather_rows": ([_vp, _vp, _vp, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_csr_sddmm": ([_vp] * 5 + [_i64, _i64, _i64, _vp], _i32),
    "cogdl_hip_edge_softmax_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_edge_softmax_fwd": ([_vp] * 3 + [_i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_edge_softmax_bwd": ([_vp] * 4 + [_i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_mhspmm_workspace_bytes": ([_i64, _i64, _i64, _i32], _sz),
    "cogdl_hip_mhspmm": ([_vp] * 5 + [_i64, _i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_mhspmm_eid": ([_vp] * 6 + [_i64, _i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_mhsddmm": ([_vp] * 5 + [_i64, _i64, _i64, _i64, _vp], _i32),
    "cogdl_hip_scatter_max_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_scatter_max_fwd": ([_vp] * 5 + [_i64, _i64, _i64, _vp, _sz, _vp], _i32),
    "cogdl_hip_scatter_max_bwd": ([_vp] * 3 + [_i64, _i64, _i64, _vp], _i32),
    "cogdl_hip_scatter_max_bwd_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_scatter_max_bwd_csc": ([_vp] * 5 + [_i64, _i64, _i64, _vp, _sz, _vp], _i32),
    "cogdl_hip_gspmm_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_gspmm": ([_vp] * 5 + [_i32, _vp, _i32, _i32, _vp, _i64, _i64, _i64, _vp, _sz, _vp], _i32),
    "cogdl_hip_gspmm_xcd_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_gspmm_xcd": ([_vp] * 4 + [_i32, _vp, _i32, _i32, _vp, _i64, _i64, _vp, _sz, _vp], _i32),
    "cogdl_hip_gspmm_edge_grad": ([_vp] * 7 + [_i32, _i32, _vp, _vp, _i64, _i64, _vp], _i32),
    "cogdl_hip_gat_fwd_workspace_bytes": ([_i64, _i64, _i64, _i32], _sz),
    "cogdl_hip_gat_fwd": ([_vp] * 5 + [_f32] + [_vp] * 3 + [_i64, _i64, _i64, _i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_gat_bwd_workspace_bytes": ([_i64, _i64, _i64, _i64, _i64, _i32], _sz),
    "cogdl_hip_gat_bwd": ([_vp] * 7 + [_f32] + [_vp] * 8 + [_sz, _i64, _i64, _i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_gat_dropout_fwd": ([_vp] * 5 + [_f32, _f32, _u64] + [_vp] * 3 + [_i64, _i64, _i64, _i64, _i32, _vp, _sz, _vp],
                                  _i32),
    "cogdl_hip_gat_dropout_bwd": ([_vp] * 8 + [_f32, _f32, _u64] + [_vp] * 8
                                  + [_sz, _i64, _i64, _i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_edge_dropout_mask": ([_i64, _i64, _f32, _u64, _vp, _vp], _i32),
    "cogdl_hip_edge_dropout_mask_host": ([_i64, _i64, _f32, _u64, _vp], _i32),
    "cogdl_hip_csr_fingerprint": ([_vp, _vp, _i64, _i64, _vp, _vp], _i32),
    "cogdl_hip_linear_fwd_f32": ([_vp] * 4 + [_i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_linear_fwd_bf16": ([_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_head_projection_fwd": ([_vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp], _i32),
    "cogdl_hip_linear_fwd_f16": ([_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _vp], _i32),
    "cogdl_hip_linear_wgrad_workspace_bytes": ([_i64, _i64, _i64], _sz),
    "cogdl_hip_linear_wgrad_f32": ([_vp] * 4 + [_i64, _i64, _i64, _vp, _sz, _vp], _i32),
    "cogdl_hip_coo2csr_index_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_sample_adj_workspace_bytes": ([_i64, _i64, _i64], _sz),
    "cogdl_hip_sample_adj": ([_vp, _vp, _i64, _vp, _i64, _i64, _i32, _u64] + [_vp] * 4 + [_i64, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_sample_adj_padded": ([_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _u64, _vp] + [_vp] * 4
                                    + [_i64, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_sample_adj_block": ([_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _u64, _vp] + [_vp] * 4
                                   + [_i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_add_remaining_self_loops_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_add_remaining_self_loops": ([_vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_coo_norm_weights_workspace_bytes": ([_i64], _sz),
    "cogdl_hip_coo_norm_weights": ([_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_block_prepare": ([_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp], _i32),
    "cogdl_hip_subgraph_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_subgraph": ([_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_gather_feature_rows": ([_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp], _i32),
    "cogdl_hip_gather_feature_rows_i32": ([_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp], _i32),
    "cogdl_hip_add_rows_at_f32": ([_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp], _i32),
    "cogdl_hip_coo2csr_index": ([_vp, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_shard_workspace_bytes": ([_i64, _i64], _sz),
    "cogdl_hip_shard_count": ([_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_shard_fill": ([_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64] + [_vp] * 8 + [_vp, _sz, _vp], _i32),
    "cogdl_hip_bfs_step": ([_vp, _vp, _i64, _vp, _i32, _vp, _vp], _i32),
    # 64-bit CSR (ABI v7); the segment table is passed by address (ctypes.addressof(Segments))
    "cogdl_hip_csr_segments": ([_vp, _i64, _i64, _i64, _vp, _vp, _vp], _i32),
    "cogdl_hip_csr_rebase_rowptr": ([_vp, _vp, _vp, _vp], _i32),
    "cogdl_hip_csr_spmm_i64_workspace_bytes": ([_vp, _i64, _i32], _sz),
    "cogdl_hip_csr_spmm_i64": ([_vp] * 6 + [_i64, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_ordered": ([_vp] * 5 + [_i64, _i64, _i64, _i32, _i32, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_spmm_i64_ordered": ([_vp] * 6 + [_i64, _i32, _vp, _vp, _sz, _vp], _i32),
    "cogdl_hip_csr_sddmm_i64": ([_vp] * 6 + [_i64, _vp], _i32),
    "cogdl_hip_csr2csc_i64_workspace_bytes": ([_vp, _i64], _sz),
    "cogdl_hip_csr2csc_i64": ([_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp], _i32),
    "cogdl_hip_gather_rows_i64": ([_vp, _vp, _vp, _i64, _i64, _i32, _vp], _i32),
}

MAX_SEGMENTS = 64  # COGDL_HIP_MAX_SEGMENTS


class Segments(ctypes.Structure):
    """cogdl_hip_segments (include/cogdl_hip.h): the row cuts of a 64-bit CSR."""
    _fields_ = [("n", ctypes.c_int32), ("row", ctypes.c_int64 * (MAX_SEGMENTS + 1)),
                ("edge", ctypes.c_int64 * (MAX_SEGMENTS + 1))]

HOST_SIGNATURES = {
    "cogdl_host_strerror": ([_i32], ctypes.c_char_p),
    "cogdl_host_coo2csr": ([_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp], _i32),
    "cogdl_host_coo2csr_index": ([_vp, _i64, _i64, _vp, _vp], _i32),
    "cogdl_host_sample_adj": ([_vp, _vp, _i64, _vp, _i64, _i64, _i32, _u64] + [_vp] * 4 + [_i64, _i64, _vp], _i32),
    "cogdl_host_sample_adj_mt": ([_vp, _vp, _i64, _vp, _i64, _i64, _i32, _u64] + [_vp] * 4 + [_i64, _i64, _vp, _i32], _i32),
    "cogdl_host_subgraph": ([_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp], _i32),
    "cogdl_host_csr_spmm_f32": ([_vp] * 5 + [_i64, _i64, _i32], _i32),
    "cogdl_host_csr_spmm_f32_i64": ([_vp] * 5 + [_i64, _i64, _i32], _i32),
}

EUNSUPPORTED = 7  # COGDL_HIP_EUNSUPPORTED
DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

_hip = None
_host = None


class BackendError(RuntimeError):
    """The HIP backend could not be loaded, or an entry point returned a non-zero status."""


def _load(path, signatures, what):
    if not os.path.exists(path):
        raise BackendError(
            "%s not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cogdl_amd/csrc`). There is no fallback path." % (what, path))
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise BackendError("failed to load %s: %s" % (path, e)) from e
    for name, (argtypes, restype) in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BackendError("%s does not export %s (stale build?)" % (path, name)) from e
        fn.argtypes, fn.restype = argtypes, restype
    return lib


def hip():
    global _hip
    if _hip is None:
        _hip = _load(HIP_LIB_PATH, HIP_SIGNATURES, "libcogdl_hip.so (HIP kernels)")
        raw_set = _hip.cogdl_hip_set_tuning

        def set_tuning(key, value):  # (workspace sizes depend on the tuning table: the memo of workspace() goes with it)
            _WS_BYTES.clear()
            return raw_set(key, value)

        _hip.cogdl_hip_set_tuning = set_tuning
        # A/B experiments without code changes: COGDL_AMD_TUNING="key=value,key=value" -> cogdl_hip_set_tuning
        for item in filter(None, os.environ.get("COGDL_AMD_TUNING", "").split(",")):
            key, _, value = item.partition("=")
            if _hip.cogdl_hip_set_tuning(int(key), int(value)) != 0:
                raise BackendError("COGDL_AMD_TUNING: no tuning key %s" % key)
    return _hip


def host():
    global _host
    if _host is None:
        _host = _load(HOST_LIB_PATH, HOST_SIGNATURES, "libcogdl_host.so (host operators)")
    return _host


def workspace(query, device, *args):
    """Long-row scratch of a row-wise operator: `query` names its *_workspace_bytes function (torch caching
    allocator: no hipMalloc per call).  Returns (tensor | None, nbytes)."""
    key = (query,) + args
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:  # (a pure function of its arguments: asked once per shape -- an epoch is launch-bound on the host)
        nbytes = getattr(hip(), query)(*[int(a) for a in args])
        if len(_WS_BYTES) < 4096:
            _WS_BYTES[key] = nbytes
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


_WS_BYTES = {}


class _NoCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _NoCtx()


def on_device(dev):
    """Context that makes `dev` the current HIP device for a launch -- a no-op object when it already is (the usual
    case: two device switches per operator call are measurable when an epoch is ~100 launches)."""
    return _NO_CTX if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


def ptr(t):
    """Raw address of a tensor's first element (None -> NULL)."""
    return None if t is None else t.data_ptr()


# (device index) -> hipStream_t as an int: ~0.3 us; building a torch.cuda.Stream object (torch.cuda.current_stream) costs
# ~5 us.  Resolved lazily: a torch build without CUDA/ROCm does not define the symbol, and the host-only paths
# (libcogdl_host, the CPU reference tests) must still import this module.
_raw_stream_fn = None
_STREAM_OBJ = {}


def _raw_stream(index):
    global _raw_stream_fn
    if _raw_stream_fn is None:
        _raw_stream_fn = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
            lambda i: torch.cuda.current_stream(i).cuda_stream)
    return _raw_stream_fn(index)


def stream_of(t):
    """The HIP stream torch would launch on for this tensor's device (current stream), as a raw handle."""
    return _raw_stream(t.device.index)


def current_stream_obj(dev):
    """torch.cuda.current_stream(dev) without rebuilding the Stream object on every call: the object is kept per device
    for as long as the raw handle of the current stream stays the same (an epoch is launch-bound on the host: three
    current_stream() calls per operator call were 15 us of its ~60)."""
    raw = _raw_stream(dev.index)
    hit = _STREAM_OBJ.get(dev.index)
    if hit is None or hit[0] != raw:
        hit = (raw, torch.cuda.current_stream(dev))
        _STREAM_OBJ[dev.index] = hit
    return hit[1]


def check(rc, what):
    if rc != 0:
        lib = hip()
        msg = lib.cogdl_hip_strerror(rc).decode()
        if rc == 4:
            msg += " (hipError_t=%d)" % lib.cogdl_hip_last_hip_error()
        raise BackendError("%s failed: %s" % (what, msg))


def check_host(rc, what):
    if rc != 0:
        raise BackendError("%s failed: %s" % (what, host().cogdl_host_strerror(rc).decode()))


def require_cuda(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise BackendError("cogdl_amd HIP operator called with a %s tensor; the HIP path has no CPU fallback"
                               % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise BackendError("tensors on different devices: %s vs %s" % (dev, t.device))
    return dev


def csr_structure(rowptr, colind):
    """The int32 CSR index arrays as the kernels (and the structure hash / transpose that read them through raw
    pointers) need them: dtype-checked, then made contiguous ONCE -- the returned tensors are the ones to launch
    with, to hash and to save for backward (a strided view would be read as if it were dense)."""
    if rowptr.dtype != torch.int32 or colind.dtype != torch.int32:
        raise BackendError("rowptr/colind must be int32 (got %s/%s)" % (rowptr.dtype, colind.dtype))
    if rowptr.dim() != 1 or colind.dim() != 1 or rowptr.numel() < 1:
        raise BackendError("rowptr/colind must be 1-D (rowptr non-empty), got shapes %s/%s"
                           % (tuple(rowptr.shape), tuple(colind.shape)))
    return rowptr.contiguous(), colind.contiguous()
