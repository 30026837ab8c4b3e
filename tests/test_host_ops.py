"""libcogdl_host.so (HIP-free operators) against the reference's golden vectors and the oracle. CPU only."""
import numpy as np
import pytest
import torch

from cogdl_amd import synth
from cogdl_amd._lib import BackendError
from cogdl_amd.operators import sample as ops
from cogdl_amd.operators.spmm import spmm_cpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_coo2csr_matches_reference_goldens(golden):
    z = golden("sampler")
    n = int(z["n"])
    rp, ci, ov = ops.coo2csr_cpu(T(z["row"]), T(z["col"]), T(z["val"]), n)
    assert np.array_equal(rp.numpy(), z["row_ptr"]) and np.array_equal(ci.numpy(), z["col_ind"])
    assert ov.numpy().tobytes() == z["out_val"].tobytes()
    rp2, perm = ops.coo2csr_cpu_index(T(z["row"]), T(z["col"]), n)
    assert np.array_equal(rp2.numpy(), z["row_ptr_index"]) and np.array_equal(perm.numpy(), z["perm"])


def test_docs_csr_and_strided_view(golden):
    z = golden("docs_csr")
    edges = torch.tensor([[0, 1], [1, 3], [2, 1], [4, 2], [0, 3]]).t()  # NON-contiguous view, as in the docs
    rp, perm = ops.coo2csr_cpu_index(edges[0], edges[1], 5)
    assert rp.tolist() == [0, 2, 3, 4, 4, 5]  # the reference returns garbage for this view (stride bug)
    assert edges[1][perm].tolist() == [1, 3, 3, 1, 2]
    assert rp.tolist() == z["row_indptr"].tolist()


def test_sample_adj_full_neighbourhood_matches_reference(golden):
    z = golden("sampler")
    got = ops.sample_adj_c(T(z["row_ptr"]), T(z["col_ind"]), T(z["seeds"]), -1, False)
    for g, name in zip(got, ("s_indptr", "s_indices", "s_nodes", "s_edges")):
        assert np.array_equal(g.numpy(), z[name]), name
    got = ops.subgraph_c(T(z["row_ptr"]), T(z["col_ind"]), T(z["sub"]))
    for g, name in zip(got, ("g_indptr", "g_indices", "g_nodes", "g_edges")):
        assert np.array_equal(g.numpy(), z[name]), name


@pytest.mark.parametrize("replace", [False, True])
@pytest.mark.parametrize("k", [1, 5, 10, 200])
def test_sample_adj_random_modes_structure(replace, k):
    g = synth.scaled(2000, 12, seed=4, norm=None)
    indptr, indices = g.rowptr.long(), g.colind.long()
    seeds = torch.randperm(2000, generator=torch.Generator().manual_seed(2))[:128]
    out_indptr, out_indices, nodes, edges = ops.sample_adj_c(indptr, indices, seeds, k, replace, seed=123)
    assert torch.equal(nodes[:128], seeds)  # seeds first (tests/datasets/test_data.py:31-39 invariant)
    assert nodes.unique().numel() == nodes.numel()  # relabelling is a bijection
    deg = indptr[seeds + 1] - indptr[seeds]
    cnt = out_indptr[1:] - out_indptr[:-1]
    want = torch.where(deg > 0, torch.full_like(deg, k), torch.zeros_like(deg)) if replace else torch.clamp(deg, max=k)
    assert torch.equal(cnt, want)
    # every sampled edge is a real edge of its seed, and new ids appear in discovery order
    seen = 128
    for i in range(128):
        e = edges[out_indptr[i]:out_indptr[i + 1]]
        assert torch.all((e >= indptr[seeds[i]]) & (e < indptr[seeds[i] + 1]))
        if not replace:
            assert e.unique().numel() == e.numel()
        assert torch.equal(nodes[out_indices[out_indptr[i]:out_indptr[i + 1]]], indices[e])
        for v in out_indices[out_indptr[i]:out_indptr[i + 1]].tolist():
            assert v <= seen
            seen = max(seen, v + 1)
    again = ops.sample_adj_c(indptr, indices, seeds, k, replace, seed=123)
    assert all(torch.equal(a, b) for a, b in zip(again, (out_indptr, out_indices, nodes, edges)))


@pytest.mark.parametrize("k,replace", [(10, False), (3, True), (-1, False), (40, False)])
def test_sample_adj_is_the_same_for_every_thread_count(monkeypatch, k, replace):
    """cogdl_host_sample_adj_mt: the picks are split over OpenMP threads, the result must not depend on how many
    (a seed row's random stream is a function of (seed, row) only); big enough that the library really splits."""
    g = synth.scaled(30000, 14, seed=9, topology="rmat", norm=None, self_loops=False)
    indptr, indices = g.rowptr.long(), g.colind.long()
    seeds = torch.randperm(30000, generator=torch.Generator().manual_seed(3))[:6000]
    want = None
    for threads in ("1", "2", "3", "8"):
        monkeypatch.setenv("COGDL_AMD_SAMPLER_THREADS", threads)
        got = ops.sample_adj_c(indptr, indices, seeds, k, replace, seed=77)
        if want is None:
            want = got
            assert got[1].numel() > 2 * 4096  # (one thread per ~4096 sampled edges: more than one is used)
        assert all(torch.equal(a, b) for a, b in zip(got, want)), threads
    monkeypatch.setenv("COGDL_AMD_SAMPLER_THREADS", "8")
    with pytest.raises(BackendError):
        bad = indices.clone()
        bad[7] = 30000  # a neighbour id outside the graph: found by whichever thread reads it
        ops.sample_adj_c(indptr, bad, torch.arange(30000), -1, False)


def test_sample_without_replacement_is_uniform():
    # one node with 20 neighbours, sample 5, many seeds: each neighbour picked ~ 25 % of the time
    indptr = torch.tensor([0, 20] + [20] * 20)
    indices = torch.arange(1, 21)
    hits = torch.zeros(21)
    trials = 4000
    for s in range(trials):
        _, _, nodes, _ = ops.sample_adj_c(indptr, indices, torch.tensor([0]), 5, False, seed=s)
        hits[nodes[1:]] += 1
    p = hits[1:] / trials
    assert torch.all((p - 0.25).abs() < 0.03), p


def test_sampler_rejects_bad_ids():
    indptr = torch.tensor([0, 1, 2])
    indices = torch.tensor([1, 0])
    with pytest.raises(BackendError):
        ops.sample_adj_c(indptr, indices, torch.tensor([5]), -1, False)
    with pytest.raises(BackendError):
        ops.coo2csr_cpu_index(torch.tensor([0, 7]), torch.tensor([0, 1]), 3)


def test_spmm_cpu_bit_exact_vs_reference(golden, oracle):
    z = golden("spmm_cpu")
    for c in sorted({k.split("_")[0] for k in z}):
        out = spmm_cpu(T(z[c + "_rowptr"]), T(z[c + "_colind"]), T(z[c + "_val"]), T(z[c + "_x"]))
        assert out.numpy().tobytes() == z[c + "_out"].tobytes()
    g = synth.scaled(20000, 14, seed=1)  # large enough to take the multi-threaded path
    x = torch.randn(g.num_nodes, 64, generator=torch.Generator().manual_seed(0))
    got = spmm_cpu(g.rowptr, g.colind, g.weight, x)
    assert got.numpy().tobytes() == oracle.csr_spmm(g.rowptr, g.colind, g.weight, x).tobytes()


@pytest.mark.parametrize("k", [1, 7, 63, 64, 65, 100, 128, 130, 200])
@pytest.mark.parametrize("weighted", [True, False], ids=["weighted", "unit-weights"])
def test_spmm_cpu_column_blocks_keep_the_reference_arithmetic(oracle, k, weighted, monkeypatch):
    """The host SpMM walks the columns in register-resident blocks of 64 (+ a plain remainder loop) and prefetches ahead:
    every width around the block size, with and without weights, one thread and several, byte for byte the oracle
    (= the reference's loop, operators/spmm/spmm_cpu.cpp:39-58: a separately rounded multiply and add per edge, CSR order)."""
    g = synth.scaled(3000, 9, seed=k, topology="rmat")  # (hub rows and empty rows)
    x = torch.randn(g.num_nodes, k, generator=torch.Generator().manual_seed(k))
    w = g.weight if weighted else None
    want = oracle.csr_spmm(g.rowptr, g.colind, g.weight if weighted else torch.ones_like(g.weight), x).tobytes()
    for threads in ("1", "5"):
        monkeypatch.setenv("COGDL_AMD_CPU_THREADS", threads)
        assert spmm_cpu(g.rowptr, g.colind, w, x).numpy().tobytes() == want
        # int64 row pointers (cogdl_host_csr_spmm_f32_i64: graphs of 2^31 edges and more, where the reference's `int` loop
        # -- spmm_cpu.cpp:24-33 -- overflows): the same arithmetic
        assert spmm_cpu(g.rowptr.long(), g.colind, w, x).numpy().tobytes() == want
