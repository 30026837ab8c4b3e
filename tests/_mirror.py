"""Test-side mirror of how CogDL CALLS the operator boundary (the reference itself is not
available on the GPU box).  Each function restates the call pattern of the cited dispatcher /
layer code with the cogdl_amd operators plugged in, so the parity tests read like CogDL's own.
"""
import torch
import torch.nn.functional as F

from cogdl_amd.operators.edge_softmax import csr_edge_softmax
from cogdl_amd.operators.mhspmm import csrmhspmm
from cogdl_amd.operators.scatter_max import scatter_max
from cogdl_amd.operators.spmm import csrspmm


class MiniGraph:
    """The slice of cogdl.data.Graph the dispatcher reads (cogdl/data/data.py:563-618)."""

    def __init__(self, row_indptr, col_indices, edge_weight=None, in_norm=None, out_norm=None, symmetric=True):
        self.row_indptr = row_indptr.long()
        self.col_indices = col_indices.long()
        self.raw_edge_weight = edge_weight
        self.in_norm, self.out_norm = in_norm, out_norm
        self._sym = symmetric

    def is_symmetric(self):
        return self._sym

    @property
    def edge_index(self):  # csr2coo, cogdl/data/data.py:305-309
        deg = self.row_indptr[1:] - self.row_indptr[:-1]
        row = torch.repeat_interleave(torch.arange(deg.numel(), device=deg.device), deg)
        return row, self.col_indices


def spmm(graph, x):
    """GPU branch of cogdl.utils.spmm_utils.spmm (utils/spmm_utils.py:98-109)."""
    if graph.out_norm is not None:
        x = graph.out_norm * x
    csr_data = graph.raw_edge_weight
    if x.dtype == torch.half and csr_data is not None:
        csr_data = csr_data.half()
    x = csrspmm(graph.row_indptr.int(), graph.col_indices.int(), x, csr_data, graph.is_symmetric(), actnn=False)
    if graph.in_norm is not None:
        x = graph.in_norm * x
    return x


def edge_softmax(graph, edge_val):
    """utils/spmm_utils.py:172-184 (GPU branch)."""
    if edge_val.dim() == 1:
        return csr_edge_softmax(graph.row_indptr.int(), edge_val.view(-1, 1)).view(-1)
    return csr_edge_softmax(graph.row_indptr.int(), edge_val)


def mh_spmm(graph, attention, h):
    """utils/spmm_utils.py:201-214 (GPU branch)."""
    if h.shape[1] > 1:
        out = csrmhspmm(graph.row_indptr.int(), graph.col_indices.int(), h, attention)
        return out.view(out.shape[0], -1)
    g = MiniGraph(graph.row_indptr, graph.col_indices, attention.view(-1), symmetric=graph.is_symmetric())
    return spmm(g, h.squeeze(1))


def gcn_layer(graph, x, W, b):
    """GCNLayer.forward (layers/gcn_layer.py:51-64), no norm/act/residual/dropout."""
    return spmm(graph, F.linear(x, W, b))


def gat_layer(graph, x, W, a_l, a_r, nhead, out_feats, alpha):
    """GATLayer.forward, unfused branch (layers/gat_layer.py:59-77)."""
    h = torch.matmul(x, W).view(-1, nhead, out_feats)
    row, col = graph.edge_index
    h_l = (a_l * h).sum(dim=-1)
    h_r = (a_r * h).sum(dim=-1)
    att = F.leaky_relu(h_l[row] + h_r[col], alpha)
    att = edge_softmax(graph, att)
    return mh_spmm(graph, att, h)


def sage_mean_layer(block, x, fc_W, fc_b):
    """MeanAggregator + SAGELayer.forward (layers/sage_layer.py:8-12,69-87); block is CSR-only so
    row_norm() sets in_norm = 1/deg (cogdl/data/data.py:226-231,248-252)."""
    deg = (block.row_indptr[1:] - block.row_indptr[:-1]).float()
    in_norm = torch.where(deg > 0, 1.0 / deg, torch.zeros_like(deg)).view(-1, 1)
    g = MiniGraph(block.row_indptr, block.col_indices, None, in_norm=in_norm, symmetric=True)
    out = spmm(g, x)
    return F.linear(torch.cat([x, out], dim=-1), fc_W, fc_b)


def sage_max_aggregate(block, x):
    """MaxAggregator (layers/sage_layer.py:21-29)."""
    return scatter_max(block.row_indptr.int(), block.col_indices.int(), x)
