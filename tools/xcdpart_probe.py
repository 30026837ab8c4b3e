#!/usr/bin/env python3
"""Round-6 probe: XCD-PARTITIONED columns.  Every XCD has a private 4 MiB L2; a 30 MB feature table does not fit one of them
but does fit the eight together -- if XCD x only ever gathers the columns with col % 8 == x.  Expressed on the UNCHANGED
csr_spmm kernel as a virtual CSR: a row of more than `split` edges becomes 8 sub-rows (its edges with col % 8 == x, in CSR
order, cut into pieces of <= T edges), shorter rows stay whole (on XCD row % 8); the virtual rows of XCD x are laid out so
that workgroup w (-> XCD w % 8 with the hardware's round-robin, tuning key 0 = 0) only holds rows of XCD w % 8.  Reported:
the launch with that layout, and the SAME virtual rows dealt to the XCDs at random (what the partition is worth), plus the
plain launch.  Reddit-shaped graph, csr_spmm F = 64, bf16 and fp32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

DEV = "cuda:0"
lib = _lib.hip()
GPB = 16


def build(rowptr, colind, split, T, shuffle, part=0):
    m = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(m, device=DEV), deg)
    col = colind.long()
    longrow = (deg > split)[row]
    # which XCD owns a column: col % 8 puts all of an XCD's gathers on addresses with equal bits 7..9 (128-byte rows) -- the
    # same L2 channels; groups of 2^part consecutive columns per XCD spread them
    cx = ((col >> part) % 8) if part >= 0 else ((col * 2654435761) >> 29) % 8
    xcd = torch.where(longrow, cx, row % 8)
    if shuffle:  # the same sub-rows, but which XCD runs them is unrelated to their columns
        xcd = (xcd + row) % 8
    key = (xcd * m + row)  # XCD-major, then row; stable: CSR order inside a sub-row
    order = torch.argsort(key, stable=True)
    ks = key[order]
    gkey, cnt = torch.unique_consecutive(ks, return_counts=True)
    n_p = (cnt + T - 1) // T
    vg = torch.repeat_interleave(torch.arange(gkey.numel(), device=DEV), n_p)
    first_v = torch.cumsum(n_p, 0) - n_p
    idx = torch.arange(vg.numel(), device=DEV) - first_v[vg]
    vlen = torch.where(idx < n_p[vg] - 1, torch.full_like(idx, T), cnt[vg] - T * (n_p[vg] - 1))
    vxcd, vrow = (gkey // m)[vg], (gkey % m)[vg]
    # per XCD: pad to a common multiple of GPB, then interleave blocks of GPB virtual rows round-robin over the XCDs
    counts = torch.bincount(vxcd, minlength=8)
    per = int((int(counts.max()) + GPB - 1) // GPB * GPB)
    V = 8 * per
    lens = torch.zeros(V, dtype=torch.long, device=DEV)
    slot_of = torch.empty(vg.numel(), dtype=torch.long, device=DEV)
    start = torch.cumsum(counts, 0) - counts
    k = torch.arange(vg.numel(), device=DEV) - start[vxcd]           # index inside its XCD's list
    slot = (k // GPB) * (8 * GPB) + vxcd * GPB + k % GPB             # block b of XCD x -> global block 8 b + x
    lens[slot] = vlen
    vrowptr = torch.zeros(V + 1, dtype=torch.long, device=DEV)
    torch.cumsum(lens, 0, out=vrowptr[1:])
    # edges must follow the slot order: edge -> its virtual row's slot
    v_of_edge = torch.repeat_interleave(torch.arange(vg.numel(), device=DEV), vlen)
    eorder = torch.argsort(slot[v_of_edge], stable=True)
    row_of_slot = torch.full((V,), -1, dtype=torch.long, device=DEV)
    row_of_slot[slot] = vrow
    return order[eorder], vrowptr.int(), row_of_slot, vg.numel(), float(counts.max()) / float(counts.float().mean())


def main():
    g = synth.reddit_like(seed=0, device=DEV, norm="sym")
    m = g.num_nodes
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        x, w = torch.randn(m, 64, device=DEV).to(dt), g.weight.to(dt)
        ref = csr_spmm_raw(g.rowptr, g.colind, w, x).float()
        t_plain = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10)
        lib.cogdl_hip_set_tuning(3, 1536)
        t_plain2 = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, w, x), 10)
        lib.cogdl_hip_set_tuning(3, 1024)
        print("%s plain %8.1f us (long grid 1536: %8.1f)" % (name, t_plain * 1e3, t_plain2 * 1e3), flush=True)
        for split, T, part in ((64, 256, 0), (64, 256, 3), (64, 256, 5), (64, 256, 8), (64, 256, -1), (256, 512, 5)):
            res = []
            for shuffle in (False, True):
                order, vrp, row_of_slot, nv, imb = build(g.rowptr, g.colind, split, T, shuffle, part)
                ci, wv = g.colind[order].contiguous(), w[order].contiguous()
                lib.cogdl_hip_set_tuning(0, 0)  # hardware round-robin: workgroup w -> XCD w % 8
                lib.cogdl_hip_set_tuning(1, 1 << 20)  # no long-row path: every virtual row has at most T edges
                pout = csr_spmm_raw(vrp, ci, wv, x)
                t = timeit(lambda: csr_spmm_raw(vrp, ci, wv, x), 10)
                lib.cogdl_hip_set_tuning(0, 32)
                lib.cogdl_hip_set_tuning(1, 0)
                ok = row_of_slot >= 0
                full = torch.zeros(m, 64, device=DEV).index_add_(0, row_of_slot[ok], pout[ok].float())
                err = (full - ref).abs().max().item()
                res.append((t, nv, imb, err))
            (t0, nv, imb, e0), (t1, _, _, e1) = res
            print("   part %2d split > %-4d T = %-4d  virtual rows %8d (x%.2f of rows)  XCD imbalance %.2f   partitioned %8.1f us (%.2fx)   "
                  "same rows, XCDs unrelated to columns %8.1f us   err %.2g %.2g" % (
                      part, split, T, nv, nv / m, imb, t0 * 1e3, t_plain / t0, t1 * 1e3, e0, e1), flush=True)


if __name__ == "__main__":
    main()
