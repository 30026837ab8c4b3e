"""Subset oracles for graphs too large for the CPU oracle (tests only): the oracle is run on a random SUBSET of the
output rows of A x (or of A^T g) of a multi-hundred-million-edge matrix -- every selected output row is computed in
full, from all of its edges, by oracle.csr_spmm (= cogdl/operators/spmm/spmm_cpu.cpp:24-35), and compared with the
same rows of the GPU result.  Helpers return host numpy arrays."""
import numpy as np
import torch


def rows_subset(rowptr, cols, w, rows_sel):
    """The selected rows of a CSR matrix (device or host tensors) as their own CSR: (sub_rowptr int64, cols, w), the
    edges of every row in their original order."""
    rows_sel = rows_sel.to(rowptr.device)
    start = rowptr[rows_sel].long()
    deg = rowptr[rows_sel + 1].long() - start
    sub_rowptr = torch.zeros(rows_sel.numel() + 1, dtype=torch.long, device=rowptr.device)
    torch.cumsum(deg, 0, out=sub_rowptr[1:])
    total = int(sub_rowptr[-1])
    eid = torch.repeat_interleave(start - sub_rowptr[:-1], deg) + torch.arange(total, device=rowptr.device)
    return sub_rowptr.cpu().numpy(), cols[eid].cpu().numpy(), w[eid].cpu().numpy()


def cols_subset(rowptr, cols, w, cols_sel, n_cols_total, col_offset=0):
    """All edges whose (global) column is in `cols_sel`, in CSR order (= ascending row, stable): (rows, cols, w)."""
    dev = cols.device
    mark = torch.zeros(n_cols_total, dtype=torch.bool, device=dev)
    mark[cols_sel.to(dev)] = True
    eid = torch.nonzero(mark[cols.long() + col_offset]).flatten()
    rows = torch.searchsorted(rowptr.long().contiguous(), eid, right=True) - 1
    return rows.cpu().numpy(), (cols[eid].long() + col_offset).cpu().numpy(), w[eid].cpu().numpy()


def oracle_rows(oracle, sub_rowptr, sub_cols, sub_w, fetch):
    """oracle.csr_spmm over a row subset: the columns are relabelled into the sorted-unique set of operand rows the
    subset touches, `fetch(ids int64 numpy) -> [len(ids), F] float32 numpy` supplies those rows."""
    uniq, inv = np.unique(sub_cols, return_inverse=True)
    return oracle.csr_spmm(sub_rowptr.astype(np.int32), inv.astype(np.int32), sub_w, fetch(uniq))


def oracle_cols(oracle, rows, cols, w, cols_sel, fetch):
    """(A^T g)[cols_sel] from the edge list (rows, cols, w) of the selected columns (CSR order): grouped by column with
    a stable sort -- the rows of a column ascending, the order a stable csr2csc gives -- then oracle.csr_spmm with the
    rows of g the subset touches.  Returns [len(cols_sel), F] in the order of cols_sel."""
    cols_sel = np.asarray(cols_sel)
    order_sel = np.argsort(cols_sel, kind="stable")
    sorted_sel = cols_sel[order_sel]
    k = np.searchsorted(sorted_sel, cols)  # position of every edge's column in the sorted selection
    order = np.argsort(k, kind="stable")
    counts = np.bincount(k, minlength=len(cols_sel))
    colptr = np.zeros(len(cols_sel) + 1, np.int64)
    np.cumsum(counts, out=colptr[1:])
    uniq, inv = np.unique(rows[order], return_inverse=True)
    out_sorted = oracle.csr_spmm(colptr.astype(np.int32), inv.astype(np.int32), w[order], fetch(uniq))
    out = np.empty_like(out_sorted)
    out[order_sel] = out_sorted
    return out
