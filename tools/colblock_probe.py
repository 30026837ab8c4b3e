#!/usr/bin/env python3
"""Round-6 probe (verdict item 1): does COLUMN locality pay for cache-sized feature tables?

Reddit-shaped graph (114.8 M edges, 233 k nodes), csr_spmm F = 64 bf16 / fp32: the gathered table is 30 / 60 MB -- inside the
256 MiB Infinity Cache, far beyond one XCD's 4 MiB L2 -- and the plain launch moves ~15 GB across the fabric for ~0.6 GB
compulsory.  Variants, all on the UNCHANGED kernels (the schedule is expressed as a different CSR):

  plain        one csr_spmm launch
  seq B        the verdict's probe: B column ranges as B csr_spmm_acc launches (every XCD's L2 holds the range in flight)
  virt B T     ONE launch over a VIRTUAL CSR whose rows are (column block, row, piece of <= T edges), block-major: the
               resident workgroups all gather from one column block (time locality; the slice is replicated in 8 L2s)
  xcd  B T     the same virtual rows, but every XCD walks ONE contiguous eighth of the block-major sequence (tuning key 0:
               stripe = n_rowblocks / 8): an XCD's L2 holds only the block it is in; eighths by virtual-row count
  xcdE B T     ... eighths of equal EDGE count (padded with empty virtual rows)

The virtual launches write one partial row per virtual row; the final segmented sum is priced separately (`combine`: a
torch index_add over the partial rows -- an upper bound of what a combine kernel would cost)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
GPB = 16  # rows per workgroup at F = 64 (16 lanes per row)


def virtual_rows(rowptr, colind, n_cols, B, T, bounds=None):
    """-> (order: edge permutation, vrowptr [V+1] int64, vrow_row [V] real row, vrow_blk [V])."""
    m = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(m, device=DEV), deg)
    col = colind.long()
    if bounds is None:
        width = (n_cols + B - 1) // B
        blk = col // width
    else:
        blk = torch.bucketize(col, bounds[1:-1], right=True)
    key = blk * m + row
    order = torch.argsort(key, stable=True)
    ks = key[order]
    gkey, cnt = torch.unique_consecutive(ks, return_counts=True)
    n_p = (cnt + T - 1) // T
    V = int(n_p.sum())
    vg = torch.repeat_interleave(torch.arange(gkey.numel(), device=DEV), n_p)
    first_v = torch.cumsum(n_p, 0) - n_p
    idx = torch.arange(V, device=DEV) - first_v[vg]
    vlen = torch.where(idx < n_p[vg] - 1, torch.full_like(idx, T), cnt[vg] - T * (n_p[vg] - 1))
    vrowptr = torch.zeros(V + 1, dtype=torch.long, device=DEV)
    torch.cumsum(vlen, 0, out=vrowptr[1:])
    return order, vrowptr, (gkey % m)[vg], (gkey // m)[vg]


def pad_equal_edge_eighths(vrowptr, unit):
    """Cut the virtual rows into 8 runs of (nearly) equal edge count, pad every run with empty rows to the same multiple of
    `unit` rows.  -> new vrowptr, index of every ORIGINAL virtual row in the padded sequence."""
    V = vrowptr.numel() - 1
    nnz = int(vrowptr[-1])
    cuts = torch.searchsorted(vrowptr, torch.arange(9, device=DEV) * (nnz / 8.0)).clamp_(0, V)
    cuts[0], cuts[8] = 0, V
    cuts = cuts.tolist()
    per = max(cuts[i + 1] - cuts[i] for i in range(8))
    per = (per + unit - 1) // unit * unit
    new = torch.empty(8 * per + 1, dtype=torch.long, device=DEV)
    pos = torch.empty(V, dtype=torch.long, device=DEV)
    for i in range(8):
        a, b = cuts[i], cuts[i + 1]
        new[i * per: i * per + (b - a)] = vrowptr[a:b]
        new[i * per + (b - a): (i + 1) * per] = vrowptr[b]
        pos[a:b] = torch.arange(i * per, i * per + (b - a), device=DEV)
    new[-1] = nnz
    return new, pos, per


def main():
    what = sys.argv[1:] or ["bf16", "f32"]
    t0 = torch.cuda.Event(enable_timing=True)
    g = synth.reddit_like(seed=0, device=DEV, norm="sym")
    m, nnz = g.num_nodes, g.nnz
    print("reddit-shaped: %d nodes, %d edges" % (m, nnz), flush=True)
    for name in what:
        dt = {"bf16": torch.bfloat16, "f32": torch.float32}[name]
        F = 64
        x = torch.randn(m, F, device=DEV).to(dt)
        w = g.weight.to(dt)
        ref = csr_spmm_raw(g.rowptr, g.colind, w, x)
        t_plain = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10)
        print("%s F=%d table %.1f MB   plain  %8.1f us" % (name, F, m * F * x.element_size() / 1e6, t_plain * 1e3), flush=True)
        for B in (8, 16, 32):
            # ---- the verdict's probe: B sequential launches over column ranges (real rows, long-row path as is)
            order, vrowptr, vrow, vblk = virtual_rows(g.rowptr, g.colind, m, B, 1 << 30)
            ci, wv = g.colind[order].contiguous(), w[order].contiguous()
            parts = []
            for b in range(B):
                sel = (vblk == b).nonzero().flatten()
                if sel.numel() == 0:
                    continue
                lo, hi = int(vrowptr[sel[0]]), int(vrowptr[sel[-1] + 1])
                cnt = torch.zeros(m, dtype=torch.long, device=DEV)
                cnt[vrow[sel]] = vrowptr[sel + 1] - vrowptr[sel]
                rp = torch.zeros(m + 1, dtype=torch.int32, device=DEV)
                rp[1:] = torch.cumsum(cnt, 0).int()
                parts.append((rp, ci[lo:hi], wv[lo:hi]))
            out = torch.zeros(m, F, dtype=dt, device=DEV)

            def seq():
                out.zero_()
                for rp, c, v in parts:
                    csr_spmm_raw(rp, c, v, x, out=out)
            seq()
            err = (out.float() - ref.float()).abs().max().item()
            t = timeit(seq, 10)
            print("  seq  B=%-3d                %8.1f us  (%.2fx)  max err %.3g" % (B, t * 1e3, t_plain / t, err), flush=True)
            del parts, out
            for T in (256, 512):
                order, vrowptr, vrow, vblk = virtual_rows(g.rowptr, g.colind, m, B, T)
                V = vrow.numel()
                ci, wv = g.colind[order].contiguous(), w[order].contiguous()
                rp32 = vrowptr.int()
                part = csr_spmm_raw(rp32, ci, wv, x)
                full = torch.zeros(m, F, device=DEV).index_add_(0, vrow, part.float())
                err = (full - ref.float()).abs().max().item()
                t_comb = timeit(lambda: torch.zeros(m, F, device=DEV).index_add_(0, vrow, part.float()), 5)
                t_virt = timeit(lambda: csr_spmm_raw(rp32, ci, wv, x), 10)
                nrb = (V + GPB - 1) // GPB
                lib.cogdl_hip_set_tuning(0, (nrb + 7) // 8)
                t_xcd = timeit(lambda: csr_spmm_raw(rp32, ci, wv, x), 10)
                lib.cogdl_hip_set_tuning(0, 32)
                rpE, pos, per = pad_equal_edge_eighths(vrowptr, GPB)
                rpE32 = rpE.int()
                lib.cogdl_hip_set_tuning(0, per // GPB)
                t_xcdE = timeit(lambda: csr_spmm_raw(rpE32, ci, wv, x), 10)
                partE = csr_spmm_raw(rpE32, ci, wv, x)
                lib.cogdl_hip_set_tuning(0, 32)
                errE = (torch.zeros(m, F, device=DEV).index_add_(0, vrow, partE[pos].float()) - ref.float()).abs().max().item()
                print("  virt B=%-3d T=%-4d V=%8d  block-major %8.1f us (%.2fx)   xcd-eighths %8.1f us (%.2fx)   equal-edge eighths "
                      "(%d rows) %8.1f us (%.2fx)   combine<= %7.1f us   err %.3g %.3g" % (
                          B, T, V, t_virt * 1e3, t_plain / t_virt, t_xcd * 1e3, t_plain / t_xcd, 8 * per, t_xcdE * 1e3,
                          t_plain / t_xcdE, t_comb * 1e3, err, errE), flush=True)
                del part, partE, full


if __name__ == "__main__":
    main()
