#!/usr/bin/env python3
"""BASELINE.json configs[2] end to end (a profile, not bench.py's contract line): one full-graph training step of
CogDL's GAT (cogdl/models/nn/gat.py: GATLayer(602 -> 8 x 8 heads, ELU) + GATLayer(64 -> 41 x 1 head); GATLayer.forward,
cogdl/layers/gat_layer.py:59-86) on the Reddit-shaped graph at its TRUE size (232,965 nodes, 114,848,857 nnz), in fp32
and in bf16 (configs[2]'s dtype: torch.autocast(bfloat16) around the forward, fp32 master weights -- the layer's matmul
yields bf16 features, which fused_gat_func / csrmhspmm read and write natively), through the operators a CogDL layer
would call:
  fused    attn_drop = 0:   fused_gat_func                      (gat_layer.py:68-70)
  unfused  attn_drop = 0.5 (the model's default): leaky_relu(h_l[row] + h_r[col]) -> csr_edge_softmax -> dropout ->
           csrmhspmm        (gat_layer.py:72-77; the gathers / leaky_relu / dropout are torch's)
plus the bf16 forward (inference) of the fused path.  Usage: python tools/gat_bench.py [--steps 5]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.edge_softmax import csr_edge_softmax  # noqa: E402
from cogdl_amd.operators.fused_gat import fused_gat_func, gat_forward  # noqa: E402
from cogdl_amd.operators.mhspmm import csrmhspmm  # noqa: E402

DEV = "cuda:0"


class GatLayer(torch.nn.Module):
    def __init__(self, in_feats, out_feats, nhead, attn_drop, alpha=0.2):
        super().__init__()
        self.nhead, self.out_feats, self.alpha, self.p = nhead, out_feats, alpha, attn_drop
        self.W = torch.nn.Parameter(torch.randn(in_feats, out_feats * nhead) * (1.0 / in_feats ** 0.5))
        self.a_l = torch.nn.Parameter(torch.randn(1, nhead, out_feats) * 0.1)
        self.a_r = torch.nn.Parameter(torch.randn(1, nhead, out_feats) * 0.1)

    def forward(self, g, x):
        rowptr, colind, row = g
        h = torch.matmul(x, self.W).view(-1, self.nhead, self.out_feats)
        h_l, h_r = (self.a_l * h).sum(dim=-1), (self.a_r * h).sum(dim=-1)
        if self.p == 0.0:
            out = fused_gat_func(h_l, h_r, rowptr, colind, rowptr, colind, self.alpha, h)
        else:
            att = F.leaky_relu(h_l[row] + h_r[colind.long()], self.alpha)
            att = F.dropout(csr_edge_softmax(rowptr, att), self.p, self.training)
            out = csrmhspmm(rowptr, colind, h, att)
        return out.reshape(out.shape[0], -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    n, feats, classes = 232_965, 602, 41
    gr = synth.reddit_like(seed=0, device=DEV)
    deg = (gr.rowptr[1:] - gr.rowptr[:-1]).long()
    g = (gr.rowptr, gr.colind, torch.repeat_interleave(torch.arange(n, device=DEV), deg))
    x = torch.randn(n, feats, device=DEV)
    y = torch.randint(0, classes, (n,), device=DEV)
    mask = torch.rand(n, device=DEV) < 0.66  # Reddit: 153,431 of 232,965 nodes train
    res = {"graph": {"nodes": n, "nnz": int(gr.nnz), "max_degree": int(deg.max())}}
    for name, p, amp in (("fused (attn_drop 0) f32", 0.0, None), ("fused (attn_drop 0) bf16", 0.0, torch.bfloat16),
                         ("unfused (attn_drop 0.5, model default) f32", 0.5, None),
                         ("unfused (attn_drop 0.5, model default) bf16", 0.5, torch.bfloat16)):
        torch.manual_seed(0)
        l1, l2 = GatLayer(feats, 8, 8, p).to(DEV), GatLayer(64, classes, 1, p).to(DEV)
        opt = torch.optim.Adam(list(l1.parameters()) + list(l2.parameters()), lr=0.005)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                h = F.elu(l1(g, F.dropout(x, 0.6, True)))
                out = l2(g, F.dropout(h, 0.6, True))
                loss = F.cross_entropy(out[mask].float(), y[mask])
            loss.backward()
            opt.step()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        res[name] = {"ms_per_epoch": ms, "GEdges_per_s_both_layers_fwd_bwd": 4 * gr.nnz / ms / 1e6,
                     "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
        print("%-46s %8.1f ms per full-graph training step" % (name, ms), flush=True)
        del l1, l2, opt
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    # bf16 inference forward of the aggregation (fused path), both layers' shapes
    for h, f in ((8, 8), (1, 41)):
        ar, ac = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
        feat = torch.randn(n, h, f, device=DEV).bfloat16()
        for _ in range(2):
            gat_forward(ar, ac, gr.rowptr, gr.colind, 0.2, feat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            gat_forward(ar, ac, gr.rowptr, gr.colind, 0.2, feat)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        res["fused forward bf16 H=%d F=%d" % (h, f)] = {"ms": ms, "GEdges_per_s": gr.nnz / ms / 1e6}
        print("fused forward bf16 H=%d F=%-3d                 %8.2f ms" % (h, f, ms), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
