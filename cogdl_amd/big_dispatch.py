"""Opt-in (`install(big_graphs=True)`): CogDL's `spmm` dispatcher for graphs of 2^31 edges and more.

The reference's GPU branch casts the row pointer to int32 before it calls the operator
(`fast_spmm(row_ptr.int(), col_indices.int(), ...)`, cogdl/utils/spmm_utils.py:98-109): at 2^31 edges the cast wraps and the
CUDA kernels take `int nnz` anyway -- ogbn-papers100M as `cogdl/datasets/ogb.py:50-55` preprocesses it (3.2e9 edges) cannot
run.  This front is that branch with ONE difference: for a GPU graph whose edge count reaches `BIG_EDGES` the int64 row
pointer is handed to `csrspmm` as it is (-> cogdl_amd/bigcsr.py: row segments on the 32-bit kernels), and the int32 copy
of the column ids is made once per structure instead of once per call (at 3.2e9 edges a copy is 12.8 GB).  Everything else
-- smaller graphs, CPU tensors, ActNN, GRB adjacency shortcuts -- is forwarded to the reference's own function, unchanged.
"""
import sys

import torch

from .operators.spmm import csrspmm
from .plan import tensor_key

BIG_EDGES = 2 ** 31 - 2 ** 20  # COGDL_HIP_SEGMENT_MAX_EDGES: what one 32-bit launch takes (tests lower it)
_ATTR = "__cogdl_amd_big_colind32__"
_orig = {}


def _colind32(graph):
    """graph.col_indices as int32, memoised on the Adjacency object and keyed on the identity + version of the int64 source."""
    col = graph.col_indices
    adj = graph._adj
    hit = adj.__dict__.get(_ATTR)
    key = tensor_key(col)
    if hit is None or hit[0] != key or hit[1] is not col:
        hit = (key, col, col.int())
        adj.__dict__[_ATTR] = hit
    return hit[2]


def _weights_as(graph, dtype):
    """graph.raw_edge_weight in the dtype of x, memoised like the column ids: the reference casts per call
    (`csr_data.half()`, utils/spmm_utils.py:103-104) -- 6.4 GB per call at 3.2e9 edges, and a NEW tensor identity per call,
    which would defeat every per-weights cache behind it (transposed values, the symmetry test of bigcsr.py)."""
    w = graph.raw_edge_weight
    if w is None or w.dtype == dtype:
        return w
    adj = graph._adj
    memo = adj.__dict__.setdefault(_ATTR + "w", {})
    key = (tensor_key(w), dtype)
    hit = memo.get("hit")
    if hit is None or hit[0] != key or hit[1] is not w:
        hit = (key, w, w.to(dtype))
        memo["hit"] = hit
    return hit[2]


def _edge_count(graph):
    """The edge count from a tensor SHAPE: `Adjacency.num_edges` is `row_ptr[-1]` -- a device read, i.e. a synchronisation per
    call and a capture error inside a hipGraph -- once the COO rows are dropped (cogdl/data/data.py:327-333).  `col` holds one
    entry per edge in both forms."""
    col = getattr(getattr(graph, "_adj", None), "col", None)
    return int(col.shape[0]) if torch.is_tensor(col) else 0


def make_spmm(reference_spmm):
    def spmm(graph, x, actnn=False, fast_spmm=None, fast_spmm_cpu=None):
        big = (torch.is_tensor(x) and x.is_cuda and not actnn and getattr(graph, "grb_adj", None) is None
               and _edge_count(graph) >= BIG_EDGES)
        if not big:
            return reference_spmm(graph, x, actnn=actnn, fast_spmm=fast_spmm, fast_spmm_cpu=fast_spmm_cpu)
        # the dispatcher's GPU branch, utils/spmm_utils.py:98-109, with the row pointer left as it is
        if graph.out_norm is not None:
            x = graph.out_norm * x
        csr_data = _weights_as(graph, torch.half) if x.dtype == torch.half else graph.raw_edge_weight
        x = csrspmm(graph.row_indptr, _colind32(graph), x, csr_data, graph.is_symmetric())
        if graph.in_norm is not None:
            x = graph.in_norm * x
        return x

    spmm.__doc__ = "cogdl_amd 64-bit front of cogdl.utils.spmm_utils.spmm"
    spmm._cogdl_amd_big = True
    return spmm


def install():
    su = sys.modules.get("cogdl.utils.spmm_utils")
    if su is None:
        return False
    if getattr(su.spmm, "_cogdl_amd_big", False):
        return True
    reference_spmm = su.spmm
    front = make_spmm(reference_spmm)
    for name, mod in list(sys.modules.items()):
        if name.startswith("cogdl") and mod is not None and getattr(mod, "spmm", None) is reference_spmm:
            _orig[name] = reference_spmm
            mod.spmm = front
    return True


def uninstall():
    for name, fn in _orig.items():
        mod = sys.modules.get(name)
        if mod is not None and getattr(getattr(mod, "spmm", None), "_cogdl_amd_big", False):
            mod.spmm = fn
    _orig.clear()
