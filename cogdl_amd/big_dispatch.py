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


def make_spmm(reference_spmm):
    def spmm(graph, x, actnn=False, fast_spmm=None, fast_spmm_cpu=None):
        big = (torch.is_tensor(x) and x.is_cuda and not actnn and getattr(graph, "grb_adj", None) is None
               and int(graph.num_edges) >= BIG_EDGES)
        if not big:
            return reference_spmm(graph, x, actnn=actnn, fast_spmm=fast_spmm, fast_spmm_cpu=fast_spmm_cpu)
        # the dispatcher's GPU branch, utils/spmm_utils.py:98-109, with the row pointer left as it is
        if graph.out_norm is not None:
            x = graph.out_norm * x
        csr_data = graph.raw_edge_weight
        if x.dtype == torch.half:
            csr_data = csr_data.half()
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
