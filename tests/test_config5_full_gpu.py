"""BASELINE.json configs[4] at FULL size on ONE MI355X: the papers100M-shaped graph as CogDL feeds it to GCN
(111,059,956 nodes, symmetrised: 3.2e9 edges > 2^31, cogdl/datasets/ogb.py:50-55), F = 128 fp32, through
csrspmm with 64-bit row pointers (cogdl_amd/bigcsr.py).  The oracle cannot run 3.2e9 edges in seconds (and the
reference cannot run them at all: `int` offsets, cogdl/operators/spmm/spmm_cpu.cpp:24-33), so parity is checked the
size-independent way: a few thousand SAMPLED output rows (random + the longest) are recomputed by the oracle from their
own edges -- bit-exact where the row is summed sequentially (<= the long-row threshold of its segment), 1e-5 beyond --
and grad_x on a sample of COLUMNS, whose transposed entries are found independently of the product's transpose (a scan
of colind for the sampled ids, ascending edge position = the stable transpose's order)."""
import numpy as np
import pytest
import torch

from cogdl_amd import _lib, synth
from cogdl_amd.bigcsr import clear_plans, plan_of
from cogdl_amd.operators.spmm import csrspmm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F = 128


def oracle_rows(oracle, g, rows, x):
    """The oracle's csr_spmm for the sampled `rows` of the big graph (their edges and gathered x rows, compacted)."""
    rp = g.rowptr[torch.cat([rows, rows + 1])].cpu().numpy().reshape(2, -1)
    deg = rp[1] - rp[0]
    small_rowptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(deg, out=small_rowptr[1:])
    pos = torch.cat([torch.arange(int(a), int(b), device=g.colind.device) for a, b in zip(rp[0], rp[1])])
    cols, w = g.colind[pos].long(), g.weight[pos]
    uniq, inv = torch.unique(cols, return_inverse=True)
    small = (torch.from_numpy(small_rowptr).int(), inv.int().cpu(), w.cpu(), x[uniq].cpu())
    return small, deg


def compare(oracle, got, small, deg, exact_upto):
    """Rows summed sequentially (<= the long-row threshold): bit-identical to the oracle's fp32 loop.  Longer rows are
    re-associated at piece borders -- and a hub row of 10^6 edges is where the ORACLE's own sequential fp32 sum drifts
    (~sqrt(n) eps of the terms' magnitude): those are compared with the float64 sum, within (1e-5 + 2e-7 sqrt(deg)) of
    sum |w x|."""
    want = oracle.csr_spmm(*small)
    short = deg <= exact_upto
    assert short.sum() > 0
    assert got[short].tobytes() == want[short].tobytes(), "sequentially summed rows must be bit-identical to the oracle"
    if (~short).any():
        want64, scale = oracle.csr_spmm_f64(*small), oracle.csr_spmm_abs(*small)
        tol = (1e-5 + 2e-7 * np.sqrt(deg[~short]))[:, None] * scale[~short] + 1e-6
        assert np.all(np.abs(got[~short] - want64[~short]) <= tol)


@pytest.mark.timeout(900)
def test_papers100m_full_size_symmetrised_on_one_gpu(oracle):
    free, total = torch.cuda.mem_get_info()
    if total < 250e9:
        pytest.skip("needs a 288 GB device")
    clear_plans()
    torch.cuda.empty_cache()
    g = synth.papers100m_like(DEV, symmetrise=True)
    n = g.num_nodes
    assert g.nnz == 2 * synth.PAPERS_PAIRS and g.nnz > 2 ** 31 and int(g.rowptr[-1]) == g.nnz
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(n, F, device=DEV, generator=gen).requires_grad_()
    out = csrspmm(g.rowptr, g.colind, x, g.weight, True)
    plan = plan_of(g.rowptr, g.colind, n)
    assert plan.n_segments >= 4
    seg_nnz = np.diff(plan.segment_edges())
    assert seg_nnz.max() < 2 ** 31 - 2 ** 20
    exact = min(_lib.hip().cogdl_hip_exact_row_edges(int(e)) for e in seg_nnz)

    # ---- forward: sampled rows (random, the longest, the segment borders) against the oracle
    deg_all = g.rowptr[1:] - g.rowptr[:-1]
    borders = torch.tensor([r for c in plan.segment_rows()[1:-1] for r in (c - 1, c)], device=DEV)
    rows = torch.cat([torch.randint(0, n, (3000,), device=DEV, generator=gen), torch.topk(deg_all, 2).indices, borders,
                      torch.tensor([0, n - 1], device=DEV)]).unique()
    del deg_all
    small, deg = oracle_rows(oracle, g, rows, x.detach())
    compare(oracle, out.detach()[rows].cpu().numpy(), small, deg, exact)

    # ---- backward: grad_x = A^T grad_out with grad_out := out (no fourth 57 GB tensor), on sampled columns
    gout = out.detach()
    out.backward(gout)
    del out
    cols = torch.cat([torch.randint(0, n, (1500,), device=DEV, generator=gen), torch.tensor([0, n - 1], device=DEV)]).unique()
    pos = []
    step = 1 << 28
    mark = torch.zeros(n, dtype=torch.bool, device=DEV)
    mark[cols] = True
    for lo in range(0, g.nnz, step):  # edges whose column is sampled, ascending position (= ascending row: stable order)
        hit = mark[g.colind[lo:lo + step].long()]
        pos.append(torch.nonzero(hit).flatten() + lo)
        del hit
    pos = torch.cat(pos)
    e_rows = torch.searchsorted(g.rowptr, pos, right=True) - 1
    e_cols = g.colind[pos].long()
    order = torch.sort(e_cols, stable=True).indices  # by column, ascending edge position inside a column
    pos, e_rows, e_cols = pos[order], e_rows[order], e_cols[order]
    cnt = torch.bincount(torch.searchsorted(cols, e_cols), minlength=cols.numel())
    small_rowptr = np.zeros(cols.numel() + 1, dtype=np.int64)
    np.cumsum(cnt.cpu().numpy(), out=small_rowptr[1:])
    uniq, inv = torch.unique(e_rows, return_inverse=True)
    small = (torch.from_numpy(small_rowptr).int(), inv.int().cpu(), g.weight[pos].cpu(), gout[uniq].cpu())
    t, _ = plan.transposed(g.weight)
    exact_t = min(_lib.hip().cogdl_hip_exact_row_edges(int(e)) for e in np.diff(t.segment_edges()))
    compare(oracle, x.grad[cols].cpu().numpy(), small, np.diff(small_rowptr), exact_t)
    # the transpose's structure on the same sample: column c of A = row c of A^T, sources ascending
    tp = t.rowptr[torch.cat([cols, cols + 1])].cpu().numpy().reshape(2, -1)
    assert np.array_equal(tp[1] - tp[0], np.diff(small_rowptr))
    c0 = int(cols[7])
    assert torch.equal(t.colind[int(t.rowptr[c0]):int(t.rowptr[c0 + 1])].long(), e_rows[small_rowptr[7]:small_rowptr[8]])
    del x, gout, g, plan, t
    clear_plans()
    torch.cuda.empty_cache()
