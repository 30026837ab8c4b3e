#!/usr/bin/env python3
"""GPU exploration harness (not part of the product): times every csr_spmm kernel variant, a few
roofs (copy, sequential-column gather), and the secondary ops, interleaved in one process.
Usage on the GPU box:  python tools/spmm_variants.py [--feat 128] > gpurun_out/variants.txt"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth  # noqa: E402
from cogdl_amd.operators.spmm import csr_spmm_raw, csr_sddmm_raw  # noqa: E402
from cogdl_amd.plan import csr2csc, gather_rows  # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def tune(key, value):
    from cogdl_amd import _lib

    _lib.hip().cogdl_hip_set_tuning(key, value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feat", type=int, default=128)
    args = ap.parse_args()
    dev = "cuda:0"
    print(torch.cuda.get_device_name(0))
    f = args.feat
    for topo in ("uniform", "rmat"):
        g = synth.arxiv_like(seed=0, topology=topo).to(dev)
        x = torch.randn(g.num_nodes, f, device=dev)
        balg = g.nnz * (8 + f * 4) + g.num_nodes * (4 + f * 4)
        print("== %s: N=%d nnz=%d F=%d  B_alg=%.3f GB" % (topo, g.num_nodes, g.nnz, f, balg / 1e9))
        rounds = 3
        best = {}
        for r in range(rounds):  # interleaved rounds
            for v in [-1] + list(range(16)):
                med, mn = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, v), reps=20)
                best.setdefault(v, []).append((med, mn))
        for v, lst in best.items():
            med = sorted(m for m, _ in lst)[len(lst) // 2]
            mn = min(m for _, m in lst)
            print("variant %3d  median %.1f us  min %.1f us  -> %.2f GEdges/s, %.0f GB/s alg (%.1f%% of 8 TB/s)" % (
                v, med * 1e3, mn * 1e3, g.nnz / med / 1e6, balg / med / 1e6, balg / med / 1e6 / 80))
        med, _ = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, None, x))
        print("unweighted auto: %.1f us" % (med * 1e3))
        med, _ = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, split_long_rows=False))
        print("auto, no long-row workspace (single launch): %.1f us" % (med * 1e3))
        for thr in (64, 128, 256, 512):
            tune(1, thr)
            r = [timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x, v), reps=15)[0] for v in (-1, 0)]
            print("long-thresh %4d: auto(V2L64) %.1f us, V4L32 %.1f us" % (thr, r[0] * 1e3, r[1] * 1e3))
        tune(1, 0)
        for lg in (128, 512, 1024, 2048):
            tune(3, lg)
            r = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x), reps=15)[0]
            print("long-row grid cap %5d: auto %.1f us" % (lg, r * 1e3))
        tune(3, 1024)
        # roofs
        y = torch.empty_like(x)
        med, _ = timeit(lambda: y.copy_(x))
        print("copy [N,F] fp32: %.1f us -> %.0f GB/s" % (med * 1e3, 2 * x.numel() * 4 / med / 1e6))
        seq = synth.CSRGraph(g.rowptr, (torch.arange(g.nnz, device=dev) % g.num_nodes).int(), g.weight, g.num_nodes)
        med, _ = timeit(lambda: csr_spmm_raw(seq.rowptr, seq.colind, seq.weight, x))
        print("same CSR shape, sequential columns (gather roof): %.1f us -> %.2f GEdges/s" % (med * 1e3, g.nnz / med / 1e6))
        idx = g.colind.long()
        med, _ = timeit(lambda: x.index_select(0, idx))
        print("torch index_select of nnz rows (materialised E x F): %.1f us" % (med * 1e3))
        # secondary ops
        med, _ = timeit(lambda: csr2csc(g.rowptr, g.colind, g.num_nodes), reps=10)
        print("csr2csc: %.1f us" % (med * 1e3))
        plan = csr2csc(g.rowptr, g.colind, g.num_nodes)
        med, _ = timeit(lambda: gather_rows(plan.perm, g.weight))
        print("gather weights: %.1f us" % (med * 1e3))
        med, _ = timeit(lambda: csr_sddmm_raw(g.rowptr, g.colind, x, x))
        print("sddmm F=%d: %.1f us" % (f, med * 1e3))
        for ff in (40, 64, 256):
            xx = torch.randn(g.num_nodes, ff, device=dev)
            med, _ = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, xx))
            ba = g.nnz * (8 + ff * 4) + g.num_nodes * (4 + ff * 4)
            print("auto F=%d: %.1f us -> %.2f GEdges/s, %.0f GB/s alg" % (ff, med * 1e3, g.nnz / med / 1e6, ba / med / 1e6))


def big():
    """Beyond the Infinity Cache: X = 2 GiB."""
    dev = "cuda:0"
    g = synth.scaled(4_000_000, 15, seed=1).to(dev)
    for f in (128, 64):
        x = torch.randn(g.num_nodes, f, device=dev)
        med, _ = timeit(lambda: csr_spmm_raw(g.rowptr, g.colind, g.weight, x), reps=8, warm=2)
        balg = g.nnz * (8 + f * 4) + g.num_nodes * (4 + f * 4)
        print("BIG N=%d nnz=%d F=%d: %.2f ms -> %.2f GEdges/s, %.0f GB/s alg" % (g.num_nodes, g.nnz, f, med, g.nnz / med / 1e6, balg / med / 1e6))
        del x


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big()
        sys.exit(0)
    main()
