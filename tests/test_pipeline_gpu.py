"""SURVEY.md section 8f ranks 2 and 4 on the GPU: node-induced subgraph (cogdl_hip_subgraph behind subgraph_c), the
zero-copy feature gather from pinned host memory, the overlapped sampling + gather pipeline, layer-wise full-neighbour
inference -- each against the host operator / the oracle / plain torch on the same inputs."""
import numpy as np
import pytest
import torch

from cogdl_amd import _lib, synth
from cogdl_amd.operators.sample import sample_adj_c, subgraph_c
from cogdl_amd.operators.spmm import csrspmm
from cogdl_amd.pipeline import BatchPipeline, check_gather, gather_rows_by_id, layerwise_inference, sample_blocks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, deg, seed, topology="uniform"):
    g = synth.scaled(n, deg, seed=seed, topology=topology, norm=None, self_loops=False)
    return g.rowptr.long(), g.colind.long()


@pytest.mark.parametrize("n,deg,pick", [(50, 4, 20), (3000, 12, 700), (3000, 12, 3000), (200, 0.5, 60)])
def test_subgraph_gpu_equals_host_operator_and_oracle(oracle, n, deg, pick):
    indptr, indices = _graph(n, deg, seed=n + pick)
    gen = torch.Generator().manual_seed(pick)
    node_idx = torch.randperm(n, generator=gen)[:pick]  # unsorted on purpose (keep_order=True callers)
    want = subgraph_c(indptr, indices, node_idx)  # libcogdl_host
    o = oracle.subgraph(indptr.numpy(), indices.numpy(), node_idx.numpy())
    got = subgraph_c(indptr.to(DEV), indices.to(DEV), node_idx)  # node_idx on the CPU, as Graph.csr_subgraph passes it
    for a, b, c in zip(got, want, o):
        assert a.is_cuda and torch.equal(a.cpu(), b) and np.array_equal(b.numpy(), np.asarray(c))


def test_subgraph_gpu_hub_row_duplicates_and_errors():
    g = synth.hub_csr(40, 40, hubs=((3, 5000), (7, 130)), seed=1, weighted=False)
    indptr, indices = g.rowptr.long(), g.colind.long()
    node_idx = torch.tensor([3, 7, 9, 3, 0])  # a duplicate: relabelled to its LAST position, like index_copy_ on one thread
    got = subgraph_c(indptr.to(DEV), indices.to(DEV), node_idx.to(DEV))
    want = subgraph_c(indptr, indices, node_idx)
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    with pytest.raises(_lib.BackendError):
        subgraph_c(indptr.to(DEV), indices.to(DEV), torch.tensor([1, 40]))
    empty = subgraph_c(indptr.to(DEV), indices.to(DEV), torch.zeros(0, dtype=torch.long))
    assert empty[0].tolist() == [0] and empty[1].numel() == 0


@pytest.mark.parametrize("where", ["hbm", "pinned-host"])
@pytest.mark.parametrize("dtype,f", [(torch.float32, 100), (torch.float32, 41), (torch.bfloat16, 64), (torch.float32, 1),
                                     (torch.float16, 6)])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_gather_rows_by_id(where, dtype, f, idt):
    n_src, n = 5000, 3001
    gen = torch.Generator().manual_seed(f)
    src = torch.randn(n_src, f, generator=gen).to(dtype)
    ids = torch.randint(0, n_src, (n,), generator=gen)
    src_dev = src.to(DEV) if where == "hbm" else src.pin_memory()
    out = gather_rows_by_id(src_dev, ids.to(DEV).to(idt))
    check_gather(out)
    assert out.is_cuda and torch.equal(out.cpu(), src[ids])
    ids_bad = ids.clone()
    ids_bad[17] = n_src
    with pytest.raises(_lib.BackendError):
        check_gather(gather_rows_by_id(src_dev, ids_bad.to(DEV)))
    with pytest.raises(_lib.BackendError):
        gather_rows_by_id(src, ids.to(DEV))  # pageable host memory: refused, no silent staged copy


def test_add_rows_at_distinct_ids():
    n_dst, n, k = 4000, 1500, 128
    gen = torch.Generator().manual_seed(0)
    out = torch.randn(n_dst, k, generator=gen)
    src = torch.randn(n, k, generator=gen)
    ids = torch.randperm(n_dst, generator=gen)[:n]
    want = out.clone().index_add_(0, ids, src)
    o, s_, i_ = out.to(DEV), src.to(DEV), ids.to(DEV)
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = _lib.hip().cogdl_hip_add_rows_at_f32(_lib.ptr(i_), _lib.ptr(s_), _lib.ptr(o), n, k, n_dst, _lib.ptr(bad),
                                              _lib.stream_of(o))
    _lib.check(rc, "add_rows_at")
    assert int(bad.item()) == 0 and torch.equal(o.cpu(), want)  # one add per element: bit-identical


def test_batch_pipeline_equals_unpipelined_sampling():
    indptr, indices = (t.to(DEV) for t in _graph(20000, 15, seed=5, topology="rmat"))
    n = 20000
    x_host = torch.randn(n, 32, generator=torch.Generator().manual_seed(1)).pin_memory()
    y = torch.randint(0, 7, (n,), device=DEV)
    seeds = [torch.randperm(n, generator=torch.Generator().manual_seed(100 + i))[:256].to(DEV) for i in range(5)]
    torch.manual_seed(7)  # sample_adj draws its per-call seed from torch's CPU generator
    want = []
    for s in seeds:
        n_id, adjs = sample_blocks(indptr, indices, s, [5, 5])
        want.append((n_id.cpu(), [(b[0].cpu(), b[1].cpu(), d) for (b, d) in adjs]))
    torch.manual_seed(7)
    got = 0
    for (s, n_id, adjs, xb, yb), (w_id, w_adjs), s0 in zip(BatchPipeline(indptr, indices, x_host, y, seeds, [5, 5]), want, seeds):
        # a consumer that only enqueues work (no synchronisation) between batches
        _ = (xb * 2.0).sum()
        assert torch.equal(s, s0) and torch.equal(n_id.cpu(), w_id)
        for (b, d), (wr, wc, wd) in zip(adjs, w_adjs):
            assert torch.equal(b[0].cpu(), wr) and torch.equal(b[1].cpu(), wc) and d == wd
        check_gather(xb)
        assert torch.equal(xb.cpu(), x_host[w_id]) and torch.equal(yb, y[s0])
        got += 1
    assert got == 5


def test_batch_pipeline_prepares_the_next_batch_while_the_previous_step_still_runs():
    """The overlap itself (round-2 advisor finding: the side stream used to wait for the whole main stream, which put
    sample(i+1) behind step i-1).  The consumer enqueues a LONG step (tens of ms of matmuls) and records an event behind
    it; by the time the pipeline hands out batch i it has already sampled and gathered batch i+1 (the sampler's size
    read-back has returned on the host) -- and step i-1 must still be running on the GPU at that moment."""
    indptr, indices = (t.to(DEV) for t in _graph(20000, 15, seed=5, topology="rmat"))
    n = 20000
    x = torch.randn(n, 32, device=DEV)
    seeds = [torch.randperm(n, generator=torch.Generator().manual_seed(200 + i))[:256].to(DEV) for i in range(6)]
    a = torch.randn(4096, 4096, device=DEV)

    def long_step(xb):
        b = a
        for _ in range(40):
            b = (b @ a) * 1e-3
        return b.sum() + xb.sum()

    long_step(x[:1])
    torch.cuda.synchronize()
    ends, still_running = [], []
    for i, (s, n_id, adjs, xb, yb) in enumerate(BatchPipeline(indptr, indices, x, None, seeds, [5, 5])):
        if i >= 1 and i < len(seeds) - 1:  # batch i+1 has been prepared: was step i-1 still on the GPU?
            still_running.append(not ends[i - 1].query())
        long_step(xb)
        e = torch.cuda.Event()
        e.record()
        ends.append(e)
    torch.cuda.synchronize()
    assert len(ends) == 6 and all(still_running), still_running


class _MeanConv(torch.nn.Module):
    """SAGELayer(aggr='mean') shape (cogdl/layers/sage_layer.py:8-12,69-87) over a (row_ptr, col) block."""

    def __init__(self, i, o):
        super().__init__()
        self.fc = torch.nn.Linear(2 * i, o)

    def forward(self, block, x):
        row_ptr, col = block
        deg = row_ptr[1:] - row_ptr[:-1]
        w = torch.repeat_interleave(1.0 / deg.clamp(min=1).float(), deg)
        return self.fc(torch.cat([x, csrspmm(row_ptr.int(), col.int(), x, w)], dim=-1))


@pytest.mark.parametrize("host_features", [False, True])
def test_layerwise_inference_equals_full_graph_evaluation(host_features):
    n, f = 5000, 24
    indptr, indices = _graph(n, 9, seed=3, topology="rmat")
    torch.manual_seed(0)
    convs = [_MeanConv(f, 16).to(DEV), _MeanConv(16, 5).to(DEV)]
    x = torch.randn(n, f)
    x_in = x.pin_memory() if host_features else x.to(DEV)
    out = layerwise_inference(convs, x_in, indptr.to(DEV), indices.to(DEV), batch_size=777)
    assert out.shape == (n, 5) and (out.is_cuda != host_features)
    # the same two layers over the whole graph at once, float64 on the CPU
    deg = (indptr[1:] - indptr[:-1])
    row = torch.repeat_interleave(torch.arange(n), deg)
    h = x.double()
    for i, conv in enumerate(convs):
        agg = torch.zeros(n, h.shape[1], dtype=torch.float64).index_add_(0, row, h[indices]) / deg.clamp(min=1).double().view(-1, 1)
        h = torch.cat([h, agg], 1) @ conv.fc.weight.detach().cpu().double().t() + conv.fc.bias.detach().cpu().double()
        if i == 0:
            h = torch.relu(h)
    np.testing.assert_allclose(out.cpu().numpy(), h.numpy(), rtol=2e-4, atol=2e-5)


def test_sample_adj_full_neighbourhood_gpu_equals_host_for_inference_blocks():
    indptr, indices = _graph(3000, 10, seed=2)
    batch = torch.arange(500, 900)
    got = sample_adj_c(indptr.to(DEV), indices.to(DEV), batch.to(DEV), -1, False)
    want = sample_adj_c(indptr, indices, batch, -1, False)
    assert torch.equal(got[2].cpu(), want[2]) and torch.equal(got[1].cpu(), want[1])
    assert torch.equal(got[0].cpu()[: want[0].numel()], want[0])
