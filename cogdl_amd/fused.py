"""Opt-in replacement of CogDL's `spmm` DISPATCHER (cogdl/utils/spmm_utils.py:85-123) that folds the two broadcast
multiplies it wraps around the kernel into the kernel.  `install(fused_norm=True)` rebinds
`cogdl.utils.spmm_utils.spmm` (and every module that imported it by name) to `spmm` below; everything the fused
operator does not cover -- CPU tensors, half precision, ActNN, GRB adjacency shortcuts -- is forwarded to the
reference's own function, unchanged.

Which graphs benefit: those that carry their normalisation as `out_norm` / `in_norm` vectors instead of baked-in edge
weights, i.e. CSR-only graphs (cogdl/data/data.py:240-258) -- in practice every block `Graph.sample_adj` returns, after
the `row_norm()` MeanAggregator applies (layers/sage_layer.py:8-12).
"""
import sys

import torch

from . import _lib
from . import linear as _linear
from .operators.spmm import csrspmm_fused

_orig = {}  # module name -> the reference's spmm


def make_spmm(reference_spmm):
    def spmm(graph, x, actnn=False, fast_spmm=None, fast_spmm_cpu=None):
        in_norm, out_norm = getattr(graph, "in_norm", None), getattr(graph, "out_norm", None)
        fusable = (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not actnn
                   and (in_norm is not None or out_norm is not None)
                   and getattr(graph, "grb_adj", None) is None)
        if fusable and torch.is_grad_enabled():
            # The fused operator treats edge weights and norm vectors as constants of the graph (no gradient).  Learned
            # or attention edge weights on a normalised block must keep theirs: the reference dispatcher's fast_spmm
            # (SPMMFunction) produces grad_edge_weight (operators/spmm.py:69-73), and the broadcast multiplies around
            # it carry the norms' -- so anything that requires grad goes the reference's way.
            w = getattr(graph, "raw_edge_weight", None)
            fusable = not any(torch.is_tensor(t) and t.requires_grad for t in (w, in_norm, out_norm))
        if not fusable:
            return reference_spmm(graph, x, actnn=actnn, fast_spmm=fast_spmm, fast_spmm_cpu=fast_spmm_cpu)
        return csrspmm_fused(graph.row_indptr.int(), graph.col_indices.int(), x, graph.raw_edge_weight, out_norm,
                             in_norm)

    spmm.__doc__ = "cogdl_amd fused-normalisation front of cogdl.utils.spmm_utils.spmm"
    spmm._cogdl_amd_fused = True
    return spmm


def install():
    su = sys.modules.get("cogdl.utils.spmm_utils")
    if su is None:
        return False
    if getattr(su.spmm, "_cogdl_amd_fused", False):
        return True
    reference_spmm = su.spmm
    fused = make_spmm(reference_spmm)
    for name, mod in list(sys.modules.items()):
        if name.startswith("cogdl") and mod is not None and getattr(mod, "spmm", None) is reference_spmm:
            _orig[name] = reference_spmm
            mod.spmm = fused
    return True


def uninstall():
    for name, fn in _orig.items():
        mod = sys.modules.get(name)
        if mod is not None:
            mod.spmm = fn
    _orig.clear()
    uninstall_narrow_side()
    uninstall_gat_dropout()


# ---------------------------------------------------------------------------------------------------------------------
# Aggregate on the narrower side (second half).  GCNLayer.forward is
# spmm(graph, linear(x)) (cogdl/layers/gcn_layer.py:51-53): the SpMM always runs at the OUTPUT width.  Where a layer
# widens (in_features < out_features: the first layer of the OGB example, 128 -> 256, examples/ogb/arxiv/gnn.py:148-150)
# the same result costs half the SpMM traffic the other way round:
#       A (X W^T + 1 b^T)  =  (A X) W^T + (A 1) b^T
# -- the aggregation at the input width, one extra SpMM of width 1 for the row sums of A (the graph's weights may change
# between calls, so they are not cached).  Same operator, same dispatcher, same parameters; fp32 results differ by
# reassociation only (checked to 1e-5 against the reference's order, tests/test_install_reference.py).
_orig_gcn_forward = {}


def _gcn_forward_narrow_side(self, graph, x):
    mod = sys.modules[type(self).__module__]
    spmm, lin = mod.spmm, self.linear
    if lin.in_features < lin.out_features and x.dim() == 2:
        out = torch.nn.functional.linear(spmm(graph, x), lin.weight)
        if lin.bias is not None:
            out = out + spmm(graph, torch.ones(x.shape[0], 1, dtype=x.dtype, device=x.device)) * lin.bias
    else:
        out = spmm(graph, lin(x))
    # (the rest of cogdl/layers/gcn_layer.py:55-64, unchanged)
    if self.norm is not None:
        out = self.norm(out)
    if self.act is not None:
        out = self.act(out)
    if self.residual is not None:
        out = out + self.residual(x)
    if self.dropout is not None:
        out = self.dropout(out)
    return out


def install_narrow_side():
    """Rebind cogdl.layers.gcn_layer.GCNLayer.forward to the order-choosing version (opt-in: install(narrow_side=True))."""
    mod = sys.modules.get("cogdl.layers.gcn_layer")
    if mod is None:
        return False
    cls = mod.GCNLayer
    if cls.forward is not _gcn_forward_narrow_side:
        _orig_gcn_forward[cls] = cls.forward
        cls.forward = _gcn_forward_narrow_side
    return True


def uninstall_narrow_side():
    for cls, fn in _orig_gcn_forward.items():
        cls.forward = fn
    _orig_gcn_forward.clear()


# ---------------------------------------------------------------------------------------------------------------------
# The branch of GATLayer.forward CogDL's gat model takes BY DEFAULT (attn_drop = 0.5: cogdl/models/nn/gat.py:30):
#       edge_attention = leakyrelu(h_l[row] + h_r[col]);  edge_softmax;  nn.Dropout;  mhspmm      (gat_layer.py:72-77)
# The score construction, its autograd (torch's sort-based indexing backward over an [E, H] tensor, twice per layer) and
# the dropout are the layer's own torch code: on the Reddit-shaped graph they make a training step 370 ms where the
# operators themselves take ~25.  `install(fused_gat_dropout=True)` rebinds GATLayer.forward (opt-in, like narrow_side:
# no longer the unchanged layer) so that this branch, too, is ONE fused operator: `fused_gat_dropout_func` -- the dropout
# mask a pure function of (seed, edge, head), regenerated in the backward; nothing of size [E, H] exists.  Same
# parameters, same statistics of the mask (each attention element kept with probability 1 - p and scaled by 1/(1 - p),
# independently), a different random stream than torch's; with attn_drop = 0 or in eval mode it is the reference's own
# fused branch without the `graph.is_symmetric()` restriction (the operator uses the true transpose).
_orig_gat_forward = {}


def _column_sum(t, block=512):
    """t.sum(0) for a tall [N, C] tensor in two stages ([N/block, block, C].sum(1), then .sum(0)).  torch's single-stage
    reduction over the long dimension runs on a handful of workgroups (233 k x 64 floats: 0.42 ms, the four of them 7 % of
    a GAT training step on the Reddit-shaped graph); two stages keep the chip busy (same fp32 sums, re-associated)."""
    n = t.shape[0]
    main = (n // block) * block
    if main == 0:
        return t.sum(0)
    out = t[:main].view(n // block, block, *t.shape[1:]).sum(1).sum(0)
    return out + t[main:].sum(0) if main < n else out


class _HeadProjections(torch.autograd.Function):
    """Both projections of GATLayer.forward at once: (h_l, h_r) = ((a_l * h).sum(-1), (a_r * h).sum(-1))
    (cogdl/layers/gat_layer.py:65-66), same forward arithmetic as the layer's.  The backward is where the time was (Reddit-
    shaped graph, bf16 step: ten column-sum launches, eight bf16 -> fp32 copies and the [N, H, F] products between them, 0.5 ms
    of an 11.6 ms step): the parameter gradients  grad_a[h, f] = sum_v g[v, h] feat[v, h, f]  are the diagonal blocks of the
    tall-skinny product  [g_l | g_r]^T . feat  -- ONE launch of the fp32 split-K MFMA reduction (cogdl_hip_linear_wgrad_f32,
    fixed order) -- and grad_feat = g_l a_l + g_r a_r is one fp32 expression rounded once, instead of two rounded halves
    summed by autograd."""

    @staticmethod
    def forward(ctx, a_l, a_r, feat):
        ctx.save_for_backward(a_l, a_r, feat)
        if (feat.is_cuda and feat.dim() == 3 and feat.dtype in _lib.DTYPE_CODE and a_l.dtype == torch.float32
                and a_r.dtype == torch.float32 and a_l.shape == (1,) + tuple(feat.shape[1:]) and a_r.shape == a_l.shape):
            # both projections in one pass over feat (cogdl_hip_head_projection_fwd): torch's form is two broadcast products
            # into [N, H, F] fp32 temporaries and two reductions -- 0.24 ms of an 11.2 ms GAT step on the Reddit-shaped graph
            f3 = feat.contiguous()
            n, h, f = f3.shape
            h_l = torch.empty((n, h), dtype=torch.float32, device=feat.device)
            h_r = torch.empty_like(h_l)
            with _lib.on_device(feat.device):
                rc = _lib.hip().cogdl_hip_head_projection_fwd(_lib.ptr(f3), _lib.DTYPE_CODE[f3.dtype], _lib.ptr(a_l.detach().contiguous()),
                                                              _lib.ptr(a_r.detach().contiguous()), _lib.ptr(h_l), _lib.ptr(h_r), n, h, f,
                                                              _lib.stream_of(f3))
            _lib.check(rc, "head_projection_fwd")
            return h_l, h_r
        return (a_l * feat).sum(dim=-1), (a_r * feat).sum(dim=-1)

    @staticmethod
    def backward(ctx, g_l, g_r):
        a_l, a_r, feat = ctx.saved_tensors
        n, h, f = feat.shape
        grad_feat = grad_al = grad_ar = None
        if g_l is None and g_r is None:
            return None, None, None
        g_l = torch.zeros_like(g_r) if g_l is None else g_l  # (an output the caller never used)
        g_r = torch.zeros_like(g_l) if g_r is None else g_r
        if ctx.needs_input_grad[2]:
            grad_feat = (g_l.unsqueeze(-1) * a_l + g_r.unsqueeze(-1) * a_r).to(feat.dtype)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            if feat.is_cuda and n >= _linear.MIN_ROWS and h * f <= _linear.MAX_FEATURES:
                g_cat = torch.cat([g_l.float(), g_r.float()], 1)                                # [N, 2 H]
                gw, _ = _linear.linear_wgrad(feat.reshape(n, h * f).float(), g_cat, want_bias=False)  # [2 H, H F]
                idx = torch.arange(h, device=feat.device)
                blocks = gw.view(2, h, h, f)[:, idx, idx]                                       # [2, H, F]: the diagonal blocks
                grad_al, grad_ar = blocks[0].unsqueeze(0).to(a_l.dtype), blocks[1].unsqueeze(0).to(a_r.dtype)
            else:
                ff = feat.float()
                grad_al = _column_sum((g_l.unsqueeze(-1).float() * ff).view(n, -1)).view(1, h, f).to(a_l.dtype)
                grad_ar = _column_sum((g_r.unsqueeze(-1).float() * ff).view(n, -1)).view(1, h, f).to(a_r.dtype)
        return grad_al, grad_ar, grad_feat


def _gat_forward_fused_dropout(self, graph, x):
    if not (torch.is_tensor(x) and x.is_cuda):
        return _orig_gat_forward[type(self)](self, graph, x)
    from .operators.fused_gat import fused_gat_dropout_func

    # (bf16 autocast, tall x: the hand-written MFMA product that reads x once, cogdl_amd/linear.py: matmul; else torch.matmul)
    h = _linear.matmul(x, self.W).view(-1, self.nhead, self.out_features)
    h[torch.isnan(h)] = 0.0
    h_l, h_r = _HeadProjections.apply(self.a_l, self.a_r, h)
    p = float(self.dropout.p) if self.training else 0.0
    out = fused_gat_dropout_func(h_l, h_r, graph.row_indptr.int(), graph.col_indices.int(), self.alpha, h, p)
    out = out.view(out.shape[0], -1)
    # (the rest of cogdl/layers/gat_layer.py:79-86, unchanged)
    if self.residual:
        res = self.residual(x)
        out += res
    if self.norm is not None:
        out = self.norm(out)
    if self.act is not None:
        out = self.act(out)
    return out


def install_gat_dropout():
    """Rebind cogdl.layers.gat_layer.GATLayer.forward to the fused attention-dropout version (install(fused_gat_dropout=True))."""
    mod = sys.modules.get("cogdl.layers.gat_layer")
    if mod is None:
        return False
    cls = mod.GATLayer
    if cls.forward is not _gat_forward_fused_dropout:
        _orig_gat_forward[cls] = cls.forward
        cls.forward = _gat_forward_fused_dropout
    return True


def uninstall_gat_dropout():
    for cls, fn in _orig_gat_forward.items():
        cls.forward = fn
    _orig_gat_forward.clear()
