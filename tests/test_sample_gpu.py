"""GPU neighbour sampler (cogdl_hip_sample_adj through cogdl_amd.operators.sample.sample_adj_c with CUDA tensors).
Deterministic modes: index-exact against the reference's golden output and the oracle restatement of
sample.cpp:6-144; random modes: the structural contract (seeds first, discovery-order ids, real edges, distinctness),
reproducibility per seed, uniformity."""
import numpy as np
import pytest
import torch

from cogdl_amd import synth
from cogdl_amd._lib import BackendError
from cogdl_amd.operators import sample as ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


@pytest.fixture(autouse=True, params=[0, 1], ids=["relabel-hash", "relabel-sort"])
def relabel_algo(request):
    """Both relabelling forms of csrc/sample.hip (tuning key 11): the hash table of first positions (default, 3-4
    launches) and the sort-based pipeline of rounds 1-2 (~25 launches; its sort is the library's radix transpose since round 5, rocPRIM's before)."""
    from cogdl_amd import _lib

    _lib.hip().cogdl_hip_set_tuning(11, request.param)
    yield request.param
    _lib.hip().cogdl_hip_set_tuning(11, 0)


@pytest.mark.parametrize("n,deg,batch,k,replace", [(3000, 9, 100, 5, False), (60000, 25, 1500, 10, False),
                                                   (60000, 25, 1500, 10, True), (200000, 12, 30000, 10, False),
                                                   (5000, 40, 2000, -1, False), (50, 2, 50, 3, False)])
def test_hash_relabel_equals_sort_relabel(n, deg, batch, k, replace):
    """The two relabelling forms give identical outputs (one block of Q, several blocks, the whole-row mode, duplicates
    galore at the small sizes), for the plain and for the fixed-capacity entry point."""
    from cogdl_amd import _lib

    g = synth.scaled(n, deg, seed=n + batch, norm=None, topology="rmat")
    indptr, indices = g.rowptr.long().to(DEV), g.colind.long().to(DEV)
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(3))[:batch].to(DEV)
    outs = []
    for algo in (0, 1):
        _lib.hip().cogdl_hip_set_tuning(11, algo)
        plain = ops.sample_adj_c(indptr, indices, seeds, k, replace, seed=77)
        padded = ops.sample_adj_padded(indptr, indices, seeds, k, replace, seed=77) if k >= 0 else ()
        half = (ops.sample_adj_padded(indptr, indices, seeds, k, replace, seed=77,
                                      count=torch.tensor([batch // 2], device=DEV)) if k >= 0 else ())
        outs.append([t.cpu() for t in plain + tuple(padded) + tuple(half)])
    assert len(outs[0]) == len(outs[1])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_full_neighbourhood_equals_reference_golden(golden):
    z = golden("sampler")  # produced by the reference's sample_adj_c(..., -1, False)
    rp, ci, nodes, edges = ops.sample_adj_c(T(z["row_ptr"]), T(z["col_ind"]), T(z["seeds"]), -1, False)
    b = z["seeds"].shape[0]
    assert np.array_equal(rp[:b + 1].cpu().numpy(), z["s_indptr"])
    assert rp.numel() == nodes.numel() + 1 and bool((rp[b:] == rp[b]).all())  # padded like data.py:828-830
    assert np.array_equal(ci.cpu().numpy(), z["s_indices"])
    assert np.array_equal(nodes.cpu().numpy(), z["s_nodes"])
    assert np.array_equal(edges.cpu().numpy(), z["s_edges"])


@pytest.mark.parametrize("n,deg,batch,k", [(2000, 12, 128, -1), (2000, 12, 128, 1000), (50000, 30, 1024, -1),
                                           (300, 3, 300, -1), (40, 0, 7, -1)])
def test_deterministic_modes_equal_oracle(oracle, n, deg, batch, k):
    g = synth.random_csr(n, n, deg, seed=n + batch, weighted=False)
    indptr, indices = g.rowptr.long(), g.colind.long()
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:batch]
    want = oracle.sample_adj(indptr.numpy(), indices.numpy(), seeds.numpy(), -1, False)
    got = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), seeds.to(DEV), k, False)  # k >= every degree == all
    assert np.array_equal(got[0][:batch + 1].cpu().numpy(), want[0])
    for a, w in zip(got[1:], want[1:]):
        assert np.array_equal(a.cpu().numpy(), w)


@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("k", [1, 5, 10, 70, 200])
def test_random_modes_structure(replace, k):
    g = synth.scaled(2000, 12, seed=4, norm=None)
    indptr, indices = g.rowptr.long(), g.colind.long()
    seeds = torch.randperm(2000, generator=torch.Generator().manual_seed(2))[:128]
    out = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), seeds, k, replace, seed=123)  # CPU batch, GPU graph
    assert all(t.is_cuda for t in out)
    out_indptr, out_indices, nodes, edges = (t.cpu() for t in out)
    assert torch.equal(nodes[:128], seeds)
    assert nodes.unique().numel() == nodes.numel()
    assert out_indptr.numel() == nodes.numel() + 1
    deg = indptr[seeds + 1] - indptr[seeds]
    cnt = out_indptr[1:129] - out_indptr[:128]
    want = torch.where(deg > 0, torch.full_like(deg, k), torch.zeros_like(deg)) if replace else torch.clamp(deg, max=k)
    assert torch.equal(cnt, want)
    seen = 128
    for i in range(128):
        e = edges[out_indptr[i]:out_indptr[i + 1]]
        assert torch.all((e >= indptr[seeds[i]]) & (e < indptr[seeds[i] + 1]))
        if not replace:
            assert e.unique().numel() == e.numel()
            assert torch.equal(e, torch.sort(e).values)  # ascending CSR position, like the host operator
        assert torch.equal(nodes[out_indices[out_indptr[i]:out_indptr[i + 1]]], indices[e])
        for v in out_indices[out_indptr[i]:out_indptr[i + 1]].tolist():
            assert v <= seen
            seen = max(seen, v + 1)
    assert seen == nodes.numel()
    again = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), seeds, k, replace, seed=123)
    assert all(torch.equal(a.cpu(), b) for a, b in zip(again, (out_indptr, out_indices, nodes, edges)))
    other = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), seeds, k, replace, seed=124)
    if k < int(deg.max()):
        assert not torch.equal(other[3].cpu(), edges)


def test_without_replacement_is_uniform():
    # 4000 independent seeds rows: node i (i < 4000) has the same 20 neighbours 4000..4019; sample 5 of 20
    rows, nb = 4000, 20
    indptr = torch.cat([torch.arange(0, rows * nb + 1, nb), torch.full((nb,), rows * nb)])
    indices = (torch.arange(rows * nb) % nb) + rows
    _, ci, nodes, _ = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), torch.arange(rows), 5, False, seed=7)
    picked = nodes[ci].cpu() - rows
    p = torch.bincount(picked, minlength=nb).float() / rows
    assert torch.all((p - 0.25).abs() < 0.03), p
    # hub row, k close to the degree and k > 64 (set spills over one lane each)
    indptr = torch.tensor([0, 3000] + [3000] * 3000)
    indices = torch.arange(1, 3001)
    _, ci, nodes, edges = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), torch.tensor([0]), 1000, False, seed=3)
    assert edges.unique().numel() == 1000 and int(edges.max()) < 3000
    assert 1300 < float(edges.float().mean()) < 1700


def test_with_replacement_is_uniform():
    rows, nb = 2000, 8
    indptr = torch.cat([torch.arange(0, rows * nb + 1, nb), torch.full((nb,), rows * nb)])
    indices = (torch.arange(rows * nb) % nb) + rows
    _, ci, nodes, _ = ops.sample_adj_c(indptr.to(DEV), indices.to(DEV), torch.arange(rows), 16, True, seed=11)
    p = torch.bincount(nodes[ci].cpu() - rows, minlength=nb).float() / (rows * 16)
    assert torch.all((p - 0.125).abs() < 0.01), p


def test_rejects_bad_ids_and_large_k():
    indptr = torch.tensor([0, 1, 2], device=DEV)
    indices = torch.tensor([1, 0], device=DEV)
    with pytest.raises(BackendError):
        ops.sample_adj_c(indptr, indices, torch.tensor([5]), -1, False)
    with pytest.raises(BackendError):
        ops.sample_adj_c(indptr, torch.tensor([1, 9], device=DEV), torch.tensor([1]), -1, False)
    with pytest.raises(BackendError):
        ops.sample_adj_c(indptr, indices, torch.tensor([0]), 5000, False)
    rp, ci, nodes, edges = ops.sample_adj_c(indptr, indices, torch.zeros(0, dtype=torch.long), 3, False)
    assert rp.tolist() == [0] and ci.numel() == 0 and nodes.numel() == 0


def test_sage_block_from_gpu_sampler_feeds_spmm(oracle):
    """The sampled block is a valid rectangular CSR for the aggregation kernels: mean aggregation over it equals the
    oracle's on the same block (SAGELayer's path, cogdl/layers/sage_layer.py:8-12)."""
    from cogdl_amd.operators.spmm import csrspmm

    g = synth.scaled(5000, 10, seed=9, norm=None)
    seeds = torch.randperm(5000, generator=torch.Generator().manual_seed(3))[:256]
    rp, ci, nodes, _ = ops.sample_adj_c(g.rowptr.long().to(DEV), g.colind.long().to(DEV), seeds, 10, False, seed=5)
    x = torch.randn(nodes.numel(), 32, device=DEV)
    deg = (rp[1:] - rp[:-1]).clamp(min=1).float()
    w = (1.0 / deg).repeat_interleave(rp[1:] - rp[:-1])
    out = csrspmm(rp.int(), ci.int(), x, w)
    want = oracle.csr_spmm(rp.cpu().numpy().astype(np.int32), ci.cpu().numpy().astype(np.int32), w.cpu().numpy(),
                           x.cpu().numpy())
    assert out.cpu().numpy().tobytes() == want.tobytes()
