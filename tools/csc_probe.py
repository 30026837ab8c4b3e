#!/usr/bin/env python3
"""csr2csc: the hand-written radix transpose (tuning key 10 = 2: LSD, 9-bit digits, at every size; 3: with packed intermediate
records at every size) against the default (= 0: by size): equality of the plans and kernel time, arxiv-shaped / Reddit-shaped
graphs and a sampled block.  (The MSD-first and 6-bit-digit variants of rounds 3-4 were removed in round 6: both measured
slower, profiles/r03_csr2csc_ab.txt, r04_csr2csc_ab.txt.)"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cogdl_amd import _lib, synth  # noqa: E402
from cogdl_amd.plan import csr2csc  # noqa: E402
from tools.ops_bench import timeit  # noqa: E402

lib = _lib.hip()
DEV = "cuda:0"


def graphs():
    yield "arxiv-rmat", synth.arxiv_like(seed=0, topology="rmat").to(DEV)
    yield "arxiv-uniform", synth.arxiv_like(seed=0, topology="uniform").to(DEV)
    yield "block-60k", synth.random_csr(11264, 124000, 5, seed=1, weighted=False).to(DEV)
    yield "reddit", synth.reddit_like(seed=0, device=DEV)


for name, g in graphs():
    n_cols = g.n_cols
    plans = {}
    for algo in (0, 2, 3):
        lib.cogdl_hip_set_tuning(10, algo)
        plans[algo] = csr2csc(g.rowptr, g.colind, n_cols)
        ms = timeit(lambda: csr2csc(g.rowptr, g.colind, n_cols), 10)
        nnz = g.colind.numel()
        print("%-14s nnz %10d  %-9s %9.1f us  (%5.1f %% of 16 B/edge at 8 TB/s)" % (
            name, nnz, {0: "default", 2: "radix-LSD", 3: "radix-packed"}[algo], ms * 1e3, 16 * nnz / (ms * 1e-3) / 8e12 * 100), flush=True)
    lib.cogdl_hip_set_tuning(10, 0)
    b = plans[0]
    same = all(torch.equal(a.colptr, b.colptr) and torch.equal(a.rowind, b.rowind) and torch.equal(a.perm, b.perm)
               for a in (plans[2], plans[3]))
    print("%-14s plans identical: %s" % (name, same), flush=True)

