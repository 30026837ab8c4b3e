"""BASELINE.json configs[3] at its TRUE graph size: the GPU neighbour sampler on the ogbn-products-shaped graph
(2,449,029 nodes, ~1.2e8 edges, R-MAT degrees), 8192 seeds x fan-out [10, 10] -- too many picks to replay on the CPU
oracle in test time, and random besides, so the checks are the structural invariants of sample_adj
(cogdl/operators/sample/sample.cpp:6-144; the reference's own test keeps to "seeds are a subset of the nodes",
tests/datasets/test_data.py:31-48), evaluated on the GPU for every sampled row and edge:

  * row i holds min(deg(seed_i), k) edges; its edge positions lie inside the seed's CSR row, strictly ascending
    (no replacement: no position twice);
  * col[j] is the local id of the neighbour the position points at: nodes[col[j]] == indices[edges[j]];
  * nodes starts with the seeds, holds no id twice, and every non-seed id was given out in discovery order
    (scanning col, a new id is always the running maximum + 1);
  * the fixed-capacity form returns the same block, and the two hops chain through device-side counts.
"""
import pytest
import torch

from cogdl_amd import synth
from cogdl_amd.operators.sample import sample_adj_c, sample_adj_padded
from cogdl_amd.pipeline import sample_blocks_padded

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, B, K = 2_449_029, 8192, 10


@pytest.fixture(autouse=True, params=[0, 1], ids=["relabel-hash", "relabel-sort"])
def relabel_algo(request):
    from cogdl_amd import _lib

    _lib.hip().cogdl_hip_set_tuning(11, request.param)
    yield request.param
    _lib.hip().cogdl_hip_set_tuning(11, 0)


@pytest.fixture(scope="module")
def products():
    src, dst = synth.rmat_pairs(N, int(N * 50.5 / 2), 0, device=DEV)
    g = synth.finalize(src, dst, N, norm=None, self_loops=False)
    indptr, indices = g.rowptr.long(), g.colind.long()
    assert indices.numel() > 1.1e8
    seeds = torch.randperm(N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))[:B]
    return indptr, indices, seeds


def _check_block(indptr, indices, seeds, row_ptr, col, nodes, edges, n_nodes, n_edges):
    b = seeds.numel()
    deg = indptr[seeds + 1] - indptr[seeds]
    cnt = row_ptr[1:b + 1] - row_ptr[:b]
    assert torch.equal(cnt, deg.clamp(max=K)) and int(row_ptr[b]) == n_edges
    rows = torch.repeat_interleave(torch.arange(b, device=DEV), cnt)
    e = edges[:n_edges]
    assert bool(((e >= indptr[seeds][rows]) & (e < indptr[seeds + 1][rows])).all())
    same_row = rows[1:] == rows[:-1]
    assert bool((e[1:][same_row] > e[:-1][same_row]).all())  # ascending positions: nothing drawn twice
    c = col[:n_edges]
    assert int(c.min()) >= 0 and int(c.max()) < n_nodes
    assert torch.equal(nodes[:n_nodes][c], indices[e])
    assert torch.equal(nodes[:b], seeds)
    assert torch.unique(nodes[:n_nodes]).numel() == n_nodes
    # discovery order: a new id is always the running maximum + 1 (the seeds hold 0 .. b-1)
    run = torch.cummax(torch.cat([torch.tensor([b - 1], device=DEV), c]), 0).values
    assert bool((c <= run[:-1] + 1).all()) and int(run[-1]) == n_nodes - 1


def test_products_scale_sampling_invariants_and_fixed_capacity_form(products):
    indptr, indices, seeds = products
    row_ptr, col, nodes, edges = sample_adj_c(indptr, indices, seeds, K, False, seed=11)
    n_nodes, n_edges = nodes.numel(), col.numel()
    assert n_edges > 2 * B and B < n_nodes <= B + n_edges  # (R-MAT: most random seeds have fewer than K neighbours)
    _check_block(indptr, indices, seeds, row_ptr, col, nodes, edges, n_nodes, n_edges)
    p_rp, p_col, p_nodes, p_edges, counts = sample_adj_padded(indptr, indices, seeds, K, False, seed=11)
    assert counts.tolist() == [n_nodes, n_edges, 0]
    assert torch.equal(p_rp[: n_nodes + 1], row_ptr) and torch.equal(p_col[:n_edges], col)
    assert torch.equal(p_nodes[:n_nodes], nodes) and torch.equal(p_edges[:n_edges], edges)


def test_products_scale_two_hops_chain_through_device_counts(products):
    indptr, indices, seeds = products
    n_id, adjs, counts = sample_blocks_padded(indptr, indices, seeds, [K, K], seed=5)
    (n1, e1, f1), (n2, e2, f2) = (c.tolist() for c in counts)
    assert f1 == 0 and f2 == 0 and B < n1 <= B * (1 + K) and n1 < n2 <= n_id.numel()
    (rp2, col2), dst2 = adjs[0]  # outermost hop first: its seed slots are hop 1's node slots
    (rp1, col1), dst1 = adjs[1]
    assert dst1 == B and dst2 == B * (1 + K) and n_id.numel() == dst2 * (1 + K)
    hop1_nodes = n_id[:n1]  # hop 2 relabels its seeds first: ids 0 .. n1-1 are hop 1's nodes in their order
    blk1 = sample_adj_padded(indptr, indices, seeds, K, False, seed=5)
    assert torch.equal(blk1[2][:n1], hop1_nodes)
    # hop 2 sampled exactly the n1 seed slots in use: rows beyond them are empty
    assert int(rp2[n1]) == e2 and bool((rp2[n1:] == e2).all())
    # _check_block's invariants on hop 2 need its edge positions, which sample_blocks_padded does not return: resample
    # the hop with the same seed word (hop seeds are spaced by HOP_SEED_STRIDE, cogdl_amd/pipeline.py)
    from cogdl_amd.pipeline import HOP_SEED_STRIDE

    r_rp, r_col, r_nodes, r_edges, r_cnt = sample_adj_padded(indptr, indices, blk1[2], K, False,
                                                            seed=(5 + HOP_SEED_STRIDE) % (1 << 64), count=blk1[4][0:1])
    assert r_cnt.tolist() == [n2, e2, 0] and torch.equal(r_col, col2) and torch.equal(r_nodes, n_id)
    _check_block(indptr, indices, hop1_nodes, r_rp, r_col, r_nodes, r_edges, n2, e2)


def test_products_scale_hub_row_picks_are_uniform(products):
    """Uniformity where the RNG streams are most likely to show bias: the HUB row of the products-shaped graph (R-MAT:
    ~1e5 neighbours).  400 independent draws of 64 of its edges (different seed words), chi-square of the 25,600 picked
    positions against the uniform distribution -- over 128 contiguous ranges of the row (a biased range reduction of the
    32-bit draws would tilt them) and over the position modulo 64 (a per-lane stream defect would)."""
    indptr, indices, _ = products
    deg = indptr[1:] - indptr[:-1]
    hub = int(torch.argmax(deg))
    d = int(deg[hub])
    assert d > 30_000, d
    k, draws = 64, 400
    seed_node = torch.tensor([hub], device=DEV)
    picks = []
    for t in range(draws):
        _, col, nodes, edges = sample_adj_c(indptr, indices, seed_node, k, False, seed=1000 + t)
        assert edges.numel() == k and torch.unique(edges).numel() == k
        picks.append(edges - indptr[hub])
    pos = torch.cat(picks)
    assert int(pos.min()) >= 0 and int(pos.max()) < d
    every = torch.arange(d, device=DEV)
    for nb, bucket in ((128, lambda p: p * 128 // d), (64, lambda p: p % 64)):
        size = torch.bincount(bucket(every), minlength=nb).double()
        expected = size / d * pos.numel()
        got = torch.bincount(bucket(pos), minlength=nb).double()
        chi2 = float(((got - expected) ** 2 / expected).sum())
        dof = nb - 1
        assert chi2 < dof + 6 * (2 * dof) ** 0.5, (nb, chi2)  # mean dof, sd sqrt(2 dof): six sigma
