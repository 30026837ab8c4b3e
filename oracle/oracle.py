"""ctypes front-end of the CPU oracle (oracle/cogdl_oracle.c) and loader of oracle/_ref.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg -- never from cogdl_amd/ (tests/test_no_oracle_in_product.py enforces it).

All functions take/return numpy arrays (C-contiguous); torch tensors are accepted and
converted with ``.detach().cpu().numpy()``.
"""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(ref=True):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def _np(a, dtype):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(ref=False)
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in dir(_Sigs):
            if name.startswith("oracle_"):
                fn = getattr(_lib, name)
                fn.argtypes, fn.restype = getattr(_Sigs, name)
    return _lib


_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


class _Sigs:
    oracle_csr_spmm_f32 = ([_vp] * 5 + [_i64, _i64, _i32], None)
    oracle_csr_spmm_f64acc = ([_vp] * 5 + [_i64, _i64], None)
    oracle_csr_spmm_absacc = ([_vp] * 5 + [_i64, _i64], None)
    oracle_csr2csc = ([_vp] * 3 + [_i64, _i64] + [_vp] * 4, None)
    oracle_csr_sddmm = ([_vp] * 5 + [_i64, _i64], None)
    oracle_edge_softmax_fwd = ([_vp] * 3 + [_i64, _i64], None)
    oracle_edge_softmax_bwd = ([_vp] * 4 + [_i64, _i64], None)
    oracle_mhspmm = ([_vp] * 5 + [_i64] * 3, None)
    oracle_mhsddmm = ([_vp] * 5 + [_i64] * 3, None)
    oracle_mhtranspose = ([_vp] * 3 + [_i64, _i64], None)
    oracle_scatter_max_fwd = ([_vp] * 5 + [_i64, _i64, _i32], None)
    oracle_scatter_max_bwd = ([_vp] * 3 + [_i64] * 3, None)
    oracle_src_op_e_aggr = ([_vp] * 4 + [_i32, _vp, _i32, _i32, _vp, _i64, _i64, _i64], None)
    oracle_coo2csr = ([_vp] * 3 + [_i64, _i64] + [_vp] * 3, None)
    oracle_coo2csr_index = ([_vp, _i64, _i64, _vp, _vp], None)
    oracle_sample_adj = ([_vp, _vp, _i64, _vp, _i64, _i64, _i32] + [_vp] * 4 + [_i64, _i64, _vp], _i32)
    oracle_subgraph = ([_vp, _vp, _i64, _vp, _i64] + [_vp] * 3 + [_i64, _vp], _i32)
    oracle_gat_fwd = ([_vp] * 5 + [_f32] + [_vp] * 3 + [_i64] * 3, None)
    oracle_gat_bwd = ([_vp] * 7 + [_f32] + [_vp] * 9 + [_i64] * 4, None)
    oracle_num_threads = ([], _i32)


def num_threads():
    return int(lib().oracle_num_threads())


# ---------------------------------------------------------------------------- spmm family
def csr_spmm(rowptr, colind, val, dense, nthreads=1):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    val, dense = _np(val, np.float32), _np(dense, np.float32)
    m, k = rowptr.shape[0] - 1, dense.shape[1]
    out = np.empty((m, k), np.float32)
    lib().oracle_csr_spmm_f32(_p(rowptr), _p(colind), _p(val), _p(dense), _p(out), m, k, nthreads)
    return out


def csr_spmm_f64(rowptr, colind, val, dense):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    val, dense = _np(val, np.float32), _np(dense, np.float32)
    m, k = rowptr.shape[0] - 1, dense.shape[1]
    out = np.empty((m, k), np.float64)
    lib().oracle_csr_spmm_f64acc(_p(rowptr), _p(colind), _p(val), _p(dense), _p(out), m, k)
    return out


def csr_spmm_abs(rowptr, colind, val, dense):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    val, dense = _np(val, np.float32), _np(dense, np.float32)
    m, k = rowptr.shape[0] - 1, dense.shape[1]
    out = np.empty((m, k), np.float64)
    lib().oracle_csr_spmm_absacc(_p(rowptr), _p(colind), _p(val), _p(dense), _p(out), m, k)
    return out


def csr2csc(rowptr, colind, val=None, n_cols=None):
    """-> (colptr, rowind, val_t or None, perm)"""
    rowptr, colind, val = _np(rowptr, np.int32), _np(colind, np.int32), _np(val, np.float32)
    m = rowptr.shape[0] - 1
    n_cols = m if n_cols is None else int(n_cols)
    nnz = int(rowptr[m])
    colptr = np.empty(n_cols + 1, np.int32)
    rowind = np.empty(nnz, np.int32)
    val_t = np.empty(nnz, np.float32) if val is not None else None
    perm = np.empty(nnz, np.int32)
    lib().oracle_csr2csc(_p(rowptr), _p(colind), _p(val), m, n_cols, _p(colptr), _p(rowind), _p(val_t), _p(perm))
    return colptr, rowind, val_t, perm


def csr_sddmm(rowptr, colind, d1, d2):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    d1, d2 = _np(d1, np.float32), _np(d2, np.float32)
    m, k = rowptr.shape[0] - 1, d1.shape[1]
    out = np.empty(int(rowptr[m]), np.float32)
    lib().oracle_csr_sddmm(_p(rowptr), _p(colind), _p(d1), _p(d2), _p(out), m, k)
    return out


# --------------------------------------------------------------------------- edge softmax
def edge_softmax_fwd(rowptr, values):
    rowptr, values = _np(rowptr, np.int32), _np(values, np.float32)
    out = np.empty_like(values)
    lib().oracle_edge_softmax_fwd(_p(rowptr), _p(values), _p(out), rowptr.shape[0] - 1, values.shape[1])
    return out


def edge_softmax_bwd(rowptr, softmax, grad):
    rowptr, softmax, grad = _np(rowptr, np.int32), _np(softmax, np.float32), _np(grad, np.float32)
    out = np.empty_like(softmax)
    lib().oracle_edge_softmax_bwd(_p(rowptr), _p(softmax), _p(grad), _p(out), rowptr.shape[0] - 1, softmax.shape[1])
    return out


# ------------------------------------------------------------------------------ multihead
def mhspmm(rowptr, colind, att, feat):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    att, feat = _np(att, np.float32), _np(feat, np.float32)
    v, (_, h, f) = rowptr.shape[0] - 1, feat.shape
    out = np.empty((v, h, f), np.float32)
    lib().oracle_mhspmm(_p(rowptr), _p(colind), _p(att), _p(feat), _p(out), v, h, f)
    return out


def mhsddmm(rowptr, colind, grad, feat):
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    grad, feat = _np(grad, np.float32), _np(feat, np.float32)
    v, (_, h, f) = rowptr.shape[0] - 1, feat.shape
    out = np.empty((int(rowptr[v]), h), np.float32)
    lib().oracle_mhsddmm(_p(rowptr), _p(colind), _p(grad), _p(feat), _p(out), v, h, f)
    return out


def mhtranspose(perm, att):
    perm, att = _np(perm, np.int32), _np(att, np.float32)
    out = np.empty_like(att)
    lib().oracle_mhtranspose(_p(perm), _p(att), _p(out), att.shape[0], att.shape[1])
    return out


def gat_fwd(rowptr, colind, h_l, h_r, feat, slope, return_att=False, drop=None):
    """drop: nullable [E,H] values of nn.Dropout on the attention (0 / 1/(1-p)), cogdl/layers/gat_layer.py:75."""
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    h_l, h_r, feat = _np(h_l, np.float32), _np(h_r, np.float32), _np(feat, np.float32)
    drop = None if drop is None else _np(drop, np.float32)
    v, (_, h, f) = rowptr.shape[0] - 1, feat.shape
    out = np.empty((v, h, f), np.float32)
    att = np.empty((int(rowptr[v]), h), np.float32) if return_att else None
    lib().oracle_gat_fwd(_p(rowptr), _p(colind), _p(h_l), _p(h_r), _p(feat), float(slope), _p(out), _p(att), _p(drop),
                         v, h, f)
    return (out, att) if return_att else out


def gat_bwd(rowptr, colind, h_l, h_r, feat, slope, gout, n_src=None, scales=False, drop=None):
    """fp64 gradients of the unfused GAT composition -> (grad_feat [n_src,H,F], grad_h_l [V,H], grad_h_r [n_src,H]);
    scales=True appends the three sums of absolute terms (the base of a floating-point tolerance).
    drop: nullable [E,H] dropout values on the attention, in CSR edge order."""
    rowptr, colind = _np(rowptr, np.int32), _np(colind, np.int32)
    h_l, h_r, feat, gout = (_np(a, np.float32) for a in (h_l, h_r, feat, gout))
    drop = None if drop is None else _np(drop, np.float32)
    v, (ns, h, f) = rowptr.shape[0] - 1, feat.shape
    n_src = ns if n_src is None else n_src
    colptr, rowind, _, perm = csr2csc(rowptr, colind, None, n_cols=n_src)
    perm = _np(perm, np.int32)
    grad_feat = np.empty((n_src, h, f), np.float32)
    grad_l, grad_r = np.empty((v, h), np.float32), np.empty((n_src, h), np.float32)
    abs_f = np.empty_like(grad_feat) if scales else None
    abs_l = np.empty_like(grad_l) if scales else None
    abs_r = np.empty_like(grad_r) if scales else None
    lib().oracle_gat_bwd(_p(rowptr), _p(colind), _p(colptr), _p(rowind), _p(h_l), _p(h_r), _p(feat), float(slope),
                         _p(gout), _p(grad_feat), _p(grad_l), _p(grad_r), _p(abs_f), _p(abs_l), _p(abs_r), _p(drop),
                         _p(perm), v, n_src, h, f)
    return (grad_feat, grad_l, grad_r, abs_f, abs_l, abs_r) if scales else (grad_feat, grad_l, grad_r)


# ---------------------------------------------------------------------------- attention dropout mask
def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123
    philox.h): ctr [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32.  The published algorithm restated in numpy;
    pinned against Random123's known-answer vectors in tests/test_gat_dropout.py."""
    m0, m1, w0, w1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = [np.asarray(ctr, np.uint32)[..., i].copy() for i in range(4)]
    k = [np.asarray(key, np.uint32)[..., i].copy() for i in range(2)]
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = m0 * c[0].astype(np.uint64)
            p1 = m1 * c[2].astype(np.uint64)
            c = [(p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k[0], p1.astype(np.uint32),
                 (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k[1], p0.astype(np.uint32)]
            k = [k[0] + w0, k[1] + w1]
    return np.stack(c, axis=-1)


def edge_dropout_mask(nnz, heads, p, seed):
    """d[e,h] of the fused GAT attention dropout (include/cogdl_hip.h, "Attention dropout"): counter (e, 0, h // 8, 0),
    key = (seed low, seed high) -> the (h % 8)-th 16-bit piece u; keep iff u >= thresh = round(p * 65536);
    d = 65536 / (65536 - thresh) where kept, else 0.  -> float32 [nnz, heads]."""
    thresh = int(np.float32(min(max(np.float32(p) * np.float32(65536.0), 0.0), 65536.0)) + np.float32(0.5))
    scale = np.float32(0.0) if thresh >= 65536 else np.float32(65536.0) / np.float32(65536 - thresh)
    out = np.zeros((nnz, heads), np.float32)
    e = np.arange(nnz, dtype=np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)
    for b in range((heads + 7) // 8):
        ctr = np.stack([e, np.zeros_like(e), np.full_like(e, b), np.zeros_like(e)], axis=-1)
        w = philox4x32_10(ctr, np.broadcast_to(key, (nnz, 2)))
        for j in range(min(8, heads - 8 * b)):
            u = (w[:, j >> 1] >> np.uint32(16 * (j & 1))) & np.uint32(0xFFFF)
            out[:, 8 * b + j] = np.where(u >= thresh, scale, np.float32(0.0))
    return out


# ---------------------------------------------------------------------------- scatter max
def scatter_max_fwd(rowptr, colind, feat, quirk=False):
    rowptr, colind, feat = _np(rowptr, np.int32), _np(colind, np.int32), _np(feat, np.float32)
    m, k = rowptr.shape[0] - 1, feat.shape[1]
    out = np.empty((m, k), np.float32)
    idx = np.empty((m, k), np.int32)
    lib().oracle_scatter_max_fwd(_p(rowptr), _p(colind), _p(feat), _p(out), _p(idx), m, k, int(quirk))
    return out, idx


def scatter_max_bwd(grad, max_id, n_src):
    grad, max_id = _np(grad, np.float32), _np(max_id, np.int32)
    m, k = grad.shape
    out = np.empty((n_src, k), np.float32)
    lib().oracle_scatter_max_bwd(_p(grad), _p(max_id), _p(out), m, k, n_src)
    return out


# ---------------------------------------------------------------------------- message ops
GSPMM_OPS = {"add": 0, "sub": 1, "mul": 2}


def src_op_e_aggr(op1, op2, x, ef, row, col, n, w=None):
    """ops.py:43-52 over the COO list in order; x None: scatter_add(ef, row, n); ef 1-D: one scalar per edge."""
    row, col = _np(row, np.int64), _np(col, np.int64)
    ef = _np(ef, np.float32)
    x = None if x is None else _np(x, np.float32)
    w = None if w is None else _np(w, np.float32)
    ef_scalar = ef.ndim == 1 or (x is not None and ef.shape[1] == 1 and x.shape[1] != 1)
    k = x.shape[1] if x is not None else ef.shape[1]
    out = np.empty((n, k), np.float32)
    lib().oracle_src_op_e_aggr(_p(row), _p(col), _p(x), _p(ef), int(ef_scalar), _p(w), GSPMM_OPS[op1],
                               int(op2 == "mean"), _p(out), n, k, row.shape[0])
    return out


# ------------------------------------------------------------------- CSR build / sampling
def coo2csr(row, col, val, num_nodes):
    row, col, val = _np(row, np.int64), _np(col, np.int64), _np(val, np.float32)
    nnz = row.shape[0]
    row_ptr = np.empty(num_nodes + 1, np.int64)
    col_ind = np.empty(nnz, np.int64)
    out_val = np.empty(nnz, np.float32)
    lib().oracle_coo2csr(_p(row), _p(col), _p(val), nnz, num_nodes, _p(row_ptr), _p(col_ind), _p(out_val))
    return row_ptr, col_ind, out_val


def coo2csr_index(row, col, num_nodes):
    row = _np(row, np.int64)
    nnz = row.shape[0]
    row_ptr = np.empty(num_nodes + 1, np.int64)
    perm = np.empty(nnz, np.int64)
    lib().oracle_coo2csr_index(_p(row), nnz, num_nodes, _p(row_ptr), _p(perm))
    return row_ptr, perm


def sample_adj(indptr, indices, node_idx, num_neighbors, replace):
    indptr, indices, node_idx = _np(indptr, np.int64), _np(indices, np.int64), _np(node_idx, np.int64)
    n, b = indptr.shape[0] - 1, node_idx.shape[0]
    deg = indptr[node_idx + 1] - indptr[node_idx]
    cap_e = int(deg.sum()) if num_neighbors < 0 else int(b * max(num_neighbors, 0)) if replace else int(
        np.minimum(deg, num_neighbors).sum())
    cap_n = b + cap_e
    out_indptr = np.empty(b + 1, np.int64)
    out_indices = np.empty(max(cap_e, 1), np.int64)
    out_nodes = np.empty(max(cap_n, 1), np.int64)
    out_edges = np.empty(max(cap_e, 1), np.int64)
    counts = np.zeros(2, np.int64)
    rc = lib().oracle_sample_adj(_p(indptr), _p(indices), n, _p(node_idx), b, int(num_neighbors), int(bool(replace)),
                                 _p(out_indptr), _p(out_indices), _p(out_nodes), _p(out_edges), cap_e, cap_n,
                                 _p(counts))
    assert rc == 0
    nn, ne = int(counts[0]), int(counts[1])
    return out_indptr, out_indices[:ne].copy(), out_nodes[:nn].copy(), out_edges[:ne].copy()


def subgraph(indptr, indices, node_idx):
    indptr, indices, node_idx = _np(indptr, np.int64), _np(indices, np.int64), _np(node_idx, np.int64)
    n, b = indptr.shape[0] - 1, node_idx.shape[0]
    cap_e = int((indptr[node_idx + 1] - indptr[node_idx]).sum())
    out_indptr = np.empty(b + 1, np.int64)
    out_indices = np.empty(max(cap_e, 1), np.int64)
    out_edges = np.empty(max(cap_e, 1), np.int64)
    counts = np.zeros(1, np.int64)
    rc = lib().oracle_subgraph(_p(indptr), _p(indices), n, _p(node_idx), b, _p(out_indptr), _p(out_indices),
                               _p(out_edges), cap_e, _p(counts))
    assert rc == 0
    ne = int(counts[0])
    return out_indptr, out_indices[:ne].copy(), np.arange(b, dtype=np.int64), out_edges[:ne].copy()


# ---------------------------------------------------------- the reference's own C++ (_ref)
def _load_ext(name, path):
    import torch  # noqa: F401  (the extension links against libtorch)

    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_available(kind="asshipped"):
    return os.path.exists(os.path.join(_HERE, "_ref", kind, "spmm_cpu.so"))


def ref_spmm_cpu(kind="asshipped"):
    """The reference's csr_spmm_cpu (cogdl/operators/spmm/spmm_cpu.cpp) built by oracle/Makefile.
    kind: 'asshipped' (-fopenmp only, CogDL's own JIT flags) or 'O3' (-O3 -mavx2 -mfma)."""
    return _load_ext("spmm_cpu", os.path.join(_HERE, "_ref", kind, "spmm_cpu.so")).csr_spmm_cpu


def ref_sampler():
    """The reference's sampler module (cogdl/operators/sample/sample.cpp)."""
    return _load_ext("sampler", os.path.join(_HERE, "_ref", "sampler.so"))


def ref_gpu_op(name):
    """A CUDA operator of the reference built for gfx950 by the reference's own JIT recipe (oracle/Makefile `ref`,
    oracle/build_ref_gpu_ops.py): 'scatter_max' (scatter_max_fp / scatter_max_bp) or 'mhspmm' (mhspmm).  GPU box only;
    used by tests/golden/make_golden_gpu.py to produce reference outputs, never by the product."""
    return _load_ext(name, os.path.join(_HERE, "_ref", "jit", name, name + ".so"))


def ref_gpu_op_available(name):
    return os.path.exists(os.path.join(_HERE, "_ref", "jit", name, name + ".so"))
