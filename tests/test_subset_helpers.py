"""tests/_subset.py (the subset oracle the true-size configs[4] tests use) against the full oracle on a small graph."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_subset_oracle_equals_full_oracle_rows(oracle):
    from _subset import cols_subset, oracle_cols, oracle_rows, rows_subset
    from cogdl_amd import synth

    g = synth.scaled(3000, 9, seed=4, topology="rmat")
    n = g.num_nodes
    x = torch.randn(n, 16, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(n, 16, generator=torch.Generator().manual_seed(2))
    full = oracle.csr_spmm(g.rowptr, g.colind, g.weight, x)
    colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=n)
    full_g = oracle.csr_spmm(colptr, rowind, w_t, gout)
    gen = torch.Generator().manual_seed(3)
    rows_sel = torch.sort(torch.randperm(n, generator=gen)[:400]).values
    cols_sel = torch.randperm(n, generator=gen)[:400]
    sub_rowptr, sub_cols, sub_w = rows_subset(g.rowptr, g.colind.long(), g.weight, rows_sel)
    want = oracle_rows(oracle, sub_rowptr, sub_cols, sub_w, lambda ids: x.numpy()[ids])
    assert want.tobytes() == full[rows_sel.numpy()].tobytes()
    e_rows, e_cols, e_w = cols_subset(g.rowptr, g.colind.long(), g.weight, cols_sel, n)
    want_g = oracle_cols(oracle, e_rows, e_cols, e_w, cols_sel.numpy(), lambda ids: gout.numpy()[ids])
    assert want_g.tobytes() == full_g[cols_sel.numpy()].tobytes()
    assert np.isfinite(want_g).all()
