"""Fixed-capacity sampled blocks and the captured mini-batch step (BASELINE.json configs[3]):
cogdl_hip_sample_adj_padded against cogdl_hip_sample_adj, cogdl_hip_csr2csc_padded / csrspmm_block against the ordinary
operators on the trimmed block, and a whole GraphSAGE step replayed as one hipGraph against the same step run eagerly
and against the reference-shaped (unpadded, synchronising) step."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cogdl_amd import _lib, graphs, synth
from cogdl_amd.operators.sample import sample_adj_c, sample_adj_padded
from cogdl_amd.operators.spmm import csrspmm, csrspmm_block
from cogdl_amd.pipeline import CapturedMiniBatchStep, HOP_SEED_STRIDE, gather_rows_by_id, sample_blocks_padded
from cogdl_amd.plan import csr2csc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, deg, seed, topology="rmat"):
    g = synth.scaled(n, deg, seed=seed, topology=topology, norm=None, self_loops=False)
    return g.rowptr.long().to(DEV), g.colind.long().to(DEV)


@pytest.mark.parametrize("k,replace", [(10, False), (3, False), (5, True), (0, False)])
@pytest.mark.parametrize("in_use", [None, 37, 0])
def test_sample_adj_padded_prefix_equals_sample_adj_and_tails_are_benign(k, replace, in_use):
    indptr, indices = _graph(4000, 9, seed=k)
    gen = torch.Generator().manual_seed(5)
    seeds = torch.randperm(4000, generator=gen)[:100].to(DEV)
    count = None if in_use is None else torch.tensor([in_use], device=DEV)
    seed_dev = torch.tensor([41], device=DEV)
    row_ptr, col, nodes, edges, counts = sample_adj_padded(indptr, indices, seeds, k, replace, seed=1000, seed_dev=seed_dev,
                                                           count=count)
    b = 100 if in_use is None else in_use
    nn, ne, flags = counts.tolist()
    assert flags == 0
    assert row_ptr.numel() == 100 + 100 * k + 1 and col.numel() == 100 * k and nodes.numel() == 100 + 100 * k
    w_rp, w_col, w_nodes, w_edges = sample_adj_c(indptr, indices, seeds[:b], k, replace, seed=1041)
    assert nn == w_nodes.numel() and ne == w_col.numel()
    assert torch.equal(row_ptr[: nn + 1], w_rp) and torch.equal(col[:ne], w_col)
    assert torch.equal(nodes[:nn], w_nodes) and torch.equal(edges[:ne], w_edges)
    # tails: empty rows, index 0 everywhere
    assert bool((row_ptr[nn:] == ne).all()) and bool((col[ne:] == 0).all()) and bool((nodes[nn:] == 0).all())
    assert bool((edges[ne:] == 0).all())


def test_sample_adj_padded_rejects_what_has_no_fixed_capacity():
    indptr, indices = _graph(100, 4, seed=1)
    seeds = torch.arange(10, device=DEV)
    with pytest.raises(_lib.BackendError):
        sample_adj_padded(indptr, indices, seeds, -1)
    with pytest.raises(_lib.BackendError):
        sample_adj_padded(indptr.cpu(), indices.cpu(), seeds.cpu(), 3)
    with pytest.raises(_lib.BackendError):
        sample_adj_padded(indptr, indices, seeds, 3, count=torch.tensor([5], dtype=torch.int32, device=DEV))
    with pytest.raises(_lib.BackendError):
        sample_adj_padded(indptr, indices, seeds, 3, seed_dev=torch.tensor([5]))  # a host tensor
    _, _, _, _, counts = sample_adj_padded(indptr, indices, torch.tensor([3, 100], device=DEV), 3)
    assert int(counts[2]) & 1  # a seed outside the graph is flagged, not fatal on the device


@pytest.mark.parametrize("algo", [0, 2, 3], ids=["default-by-size", "radix-transpose", "radix-transpose-packed-records"])
@pytest.mark.parametrize("surplus", [0, 1, 777, 20000])
@pytest.mark.parametrize("n_cols", [500, 6000], ids=["one-radix-pass", "two-radix-passes"])
def test_csr2csc_padded_ignores_the_slots_behind_the_last_row(surplus, algo, n_cols):
    _lib.hip().cogdl_hip_set_tuning(10, algo)
    try:
        _padded_transpose_case(surplus, n_cols)
    finally:
        _lib.hip().cogdl_hip_set_tuning(10, 0)


def _padded_transpose_case(surplus, n_cols=500):
    g = synth.random_csr(300, n_cols, 7, seed=surplus)
    rp, ci = g.rowptr.to(DEV), g.colind.to(DEV)
    junk = torch.randint(0, n_cols, (surplus,), dtype=torch.int32, device=DEV)
    want = csr2csc(rp, ci, n_cols)
    got = csr2csc(rp, torch.cat([ci, junk]), n_cols, padded=True)
    nnz = ci.numel()
    assert torch.equal(got.colptr, want.colptr)
    assert torch.equal(got.rowind[:nnz], want.rowind) and torch.equal(got.perm[:nnz], want.perm)
    assert int(got.rowind.max()) < 300 and int(got.rowind.min()) >= 0  # the tail stays in range


@pytest.mark.parametrize("f", [100, 128, 47])
def test_csrspmm_block_on_a_padded_block_equals_csrspmm_on_the_trimmed_one(oracle, f):
    indptr, indices = _graph(20000, 12, seed=f)
    seeds = torch.randperm(20000, device=DEV)[:256]
    row_ptr, col, nodes, _, counts = sample_adj_padded(indptr, indices, seeds, 10, seed=f)
    nn, ne, flags = counts.tolist()
    assert flags == 0 and ne < col.numel()  # (there IS a surplus: low-degree seeds, duplicates)
    n_dst, n_src_cap = 256, nodes.numel()
    rp = row_ptr[: n_dst + 1].int()
    deg = (rp[1:] - rp[:-1])
    inv = 1.0 / deg.clamp(min=1).float()
    x = torch.randn(n_src_cap, f, device=DEV, requires_grad=True)
    out = csrspmm_block(rp, col.int(), x, None, inv)
    gr = torch.randn_like(out)
    out.backward(gr)
    # the ordinary operator on the trimmed block, weights = 1 / in-degree per edge (Graph.row_norm)
    x2 = x.detach().clone().requires_grad_()
    w = torch.repeat_interleave(inv, deg.long())
    want = csrspmm(rp, col[:ne].int(), x2, w)
    want.backward(gr)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), x2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert bool((x.grad[nn:] == 0).all())  # node slots not in use receive nothing
    # and against the CPU oracle
    ref = oracle.csr_spmm(rp.cpu(), col[:ne].int().cpu(), w.cpu(), x.detach().cpu())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_rows", [None, 100, 0])
def test_block_for_spmm_equals_the_torch_expressions(n_rows):
    from cogdl_amd.graph_build import block_for_spmm

    indptr, indices = _graph(5000, 6, seed=2)
    row_ptr, col, nodes, _, counts = sample_adj_padded(indptr, indices, torch.arange(300, device=DEV), 4, seed=9)
    rp, c32, inv = block_for_spmm(row_ptr, col, n_rows)
    m = row_ptr.numel() - 1 if n_rows is None else n_rows
    assert rp.dtype == torch.int32 and c32.dtype == torch.int32 and inv.dtype == torch.float32
    assert torch.equal(rp, row_ptr[: m + 1].int()) and torch.equal(c32, col.int())
    deg = (row_ptr[1:m + 1] - row_ptr[:m]).float()
    want = torch.pow(deg, -1)
    want[torch.isinf(want)] = 0  # Adjacency.generate_normalization("row"), cogdl/data/data.py:250-252
    assert torch.equal(inv, want)
    assert block_for_spmm(row_ptr, col, n_rows, mean=False)[2] is None
    with pytest.raises(_lib.BackendError):
        block_for_spmm(row_ptr.int(), col, n_rows)


@pytest.mark.parametrize("relabel", [0, 1], ids=["hash-relabel", "sort-relabel"])
@pytest.mark.parametrize("b,k,in_use", [(300, 4, None), (100, 10, 37), (2000, 10, None), (64, 0, None), (50, 5, 0)])
def test_sampler_writes_the_block_in_the_spmm_form_itself(b, k, in_use, relabel):
    """cogdl_hip_sample_adj_block = cogdl_hip_sample_adj_padded + cogdl_hip_block_prepare without the second launch:
    int32 row pointer of the seed rows, int32 local ids, 1 / in-degree; block_for_spmm picks it up."""
    from cogdl_amd.graph_build import block_for_spmm

    indptr, indices = _graph(6000, 7, seed=b + k)
    seeds = torch.randperm(6000, device=DEV)[:b]
    count = None if in_use is None else torch.tensor([in_use], device=DEV)
    table = torch.full((2, 3), -7, dtype=torch.long, device=DEV)
    _lib.hip().cogdl_hip_set_tuning(11, relabel)
    try:
        plain = sample_adj_padded(indptr, indices, seeds, k, seed=77, count=count)
        row_ptr, col, nodes, edges, counts = sample_adj_padded(indptr, indices, seeds, k, seed=77, count=count,
                                                               counts_out=table[1], block32=True)
    finally:
        _lib.hip().cogdl_hip_set_tuning(11, 0)
    for got, want in zip((row_ptr, col, nodes, edges, counts), plain):
        assert torch.equal(got, want)
    assert counts.data_ptr() == table[1].data_ptr() and table[0].tolist() == [-7, -7, -7]  # the caller's row, only that row
    want_rp, want_col, want_inv = block_for_spmm(plain[0], plain[1], b)  # (one block_prepare launch)
    rp, c32, inv = block_for_spmm(row_ptr, col, b)                        # (nothing launched: the sampler's tensors)
    assert rp.data_ptr() == row_ptr._cogdl_block32[1].data_ptr()
    assert torch.equal(rp, want_rp) and torch.equal(c32, want_col) and torch.equal(inv, want_inv)
    assert block_for_spmm(row_ptr, col, b, mean=False)[2] is None
    assert getattr(rp, "_cogdl_max_row_edges") == k
    # another row count is not what the sampler prepared: the ordinary conversion answers
    if b > 10:
        rp2, _, _ = block_for_spmm(row_ptr, col, 10)
        assert rp2.numel() == 11 and torch.equal(rp2, row_ptr[:11].int())


def test_csrspmm_block_with_a_row_bound_is_one_launch_and_the_same_result():
    indptr, indices = _graph(20000, 12, seed=3)
    seeds = torch.randperm(20000, device=DEV)[:512]
    row_ptr, col, nodes, _, counts = sample_adj_padded(indptr, indices, seeds, 10, seed=4, block32=True)
    from cogdl_amd.graph_build import block_for_spmm

    rp, c32, inv = block_for_spmm(row_ptr, col, 512)
    x = torch.randn(nodes.numel(), 100, device=DEV, requires_grad=True)
    out = csrspmm_block(rp, c32, x, None, inv)  # the sampler's bound (10 edges per row) travels with rp
    g = torch.randn_like(out)
    out.backward(g)
    x2 = x.detach().clone().requires_grad_()
    out2 = csrspmm_block(rp.clone(), c32, x2, None, inv)  # no bound known: long-row path planned, combine kernel launched
    out2.backward(g)
    assert torch.equal(out, out2) and torch.equal(x.grad, x2.grad)
    out3 = csrspmm_block(rp.clone(), c32, x.detach(), None, inv, max_row_edges=10 ** 6)  # a bound above the threshold: ignored
    assert torch.equal(out3, out.detach())


def test_gather_reports_a_bad_id_in_the_callers_flag_word():
    from cogdl_amd.pipeline import GATHER_BAD_ID

    x = torch.randn(100, 12, device=DEV)
    word = torch.tensor([0, 6, 0], device=DEV)  # {., flags, .}: bits already set stay
    ids = torch.tensor([3, 99, 0], device=DEV)
    out = gather_rows_by_id(x, ids, flag_word=word[1:2])
    assert torch.equal(out, x[ids]) and word.tolist() == [0, 6, 0] and not hasattr(out, "_cogdl_bad_flag")
    gather_rows_by_id(x, torch.tensor([3, 100], device=DEV), flag_word=word[1:2])
    assert word.tolist() == [0, 6 | GATHER_BAD_ID, 0]
    with pytest.raises(_lib.BackendError):
        gather_rows_by_id(x, ids, flag_word=torch.zeros(1, dtype=torch.int32, device=DEV))


def test_padded_sampler_and_transpose_replay_correctly_above_a_million_slots():
    """A captured graph must stay right at any size: the sampler's sort-based relabelling keeps rocPRIM's merge sort there
    (above 1 M keys its default is onesweep, which clears its state with hipMemsetAsync -- memset nodes do not replay
    reliably), the transpose is this library's radix sort (kernels only, no memset)."""
    n = 400000
    indptr, indices = _graph(n, 12, seed=8)
    seeds = torch.randperm(n, device=DEV)[:120000]
    sd = torch.zeros(1, dtype=torch.long, device=DEV)

    def work():
        row_ptr, col, nodes, edges, counts = sample_adj_padded(indptr, indices, seeds, 10, seed=3, seed_dev=sd)
        plan = csr2csc(row_ptr[: seeds.numel() + 1].int(), col.int(), nodes.numel(), padded=True)
        return row_ptr, col, nodes, edges, counts, plan.colptr, plan.rowind, plan.perm

    want = [t.clone() for t in work()]
    assert want[1].numel() == 1200000 and int(want[4][2]) == 0
    ne = int(want[4][1])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = work()
    poison = torch.ones(64 << 20, dtype=torch.bool, device=DEV)  # (whatever the allocator hands out next is not zero)
    del poison
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        for name, a, b in zip("row_ptr col nodes edges counts colptr".split(), got, want):
            assert torch.equal(a, b), name
        assert torch.equal(got[6][:ne], want[6][:ne]) and torch.equal(got[7][:ne], want[7][:ne])


def _models():
    from tools.sage_bench import Sage

    torch.manual_seed(0)
    a = Sage(32, 64, 7).to(DEV)
    return a, copy.deepcopy(a)


def test_captured_graphsage_step_equals_the_eager_step_and_the_reference_shaped_step():
    n, b, fan = 30000, 128, [10, 10]
    indptr, indices = _graph(n, 14, seed=3)
    gen = torch.Generator(device=DEV).manual_seed(1)
    x_all = torch.randn(n, 32, device=DEV, generator=gen)
    y_all = torch.randint(0, 7, (n,), device=DEV, generator=gen)
    order = torch.randperm(n, device=DEV, generator=gen)
    m_cap, m_eag = _models()
    m_cap.eval(), m_eag.eval()  # (dropout off: the three paths must agree number for number)

    def make_step(model, seeds_buf, seed_dev, opt):
        def step():
            n_id, adjs, counts = sample_blocks_padded(indptr, indices, seeds_buf, fan, seed=77, seed_dev=seed_dev)
            xb = gather_rows_by_id(x_all, n_id)
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(model.forward_padded(xb, adjs), y_all.index_select(0, seeds_buf))
            loss.backward()
            opt.step()
            seed_dev.add_(1)
            return loss, counts
        return step

    # ---- the reference-shaped step on the same draws: unpadded blocks (sizes read back), csrspmm + plan cache
    ref_model = copy.deepcopy(m_eag)
    seeds0 = order[:b].clone()
    batch, adjs = seeds0, []
    for hop, k in enumerate(fan):
        rp, col, nodes, _ = sample_adj_c(indptr, indices, batch, k, False, seed=(77 + hop * HOP_SEED_STRIDE) % (1 << 64))
        adjs.append(((rp, col), batch.numel()))
        batch = nodes
    ref_loss = F.cross_entropy(ref_model(x_all[batch], adjs[::-1]), y_all[seeds0])
    ref_loss.backward()

    # ---- eager padded steps
    buf_e, sd_e = order[:b].clone(), torch.zeros(1, dtype=torch.long, device=DEV)
    opt_e = torch.optim.Adam(m_eag.parameters(), lr=0.01, capturable=True)
    step_e = make_step(m_eag, buf_e, sd_e, opt_e)
    first_grads = None
    eager_losses = []
    for i in range(4):
        buf_e.copy_(order[i * b:(i + 1) * b])
        loss, counts = step_e()
        assert int(counts[0][2]) == 0 and int(counts[1][2]) == 0
        eager_losses.append(float(loss.detach()))
        if i == 0:
            first_grads = [p.grad.clone() for p in m_eag.parameters()]
    ref_loss = ref_loss.detach()
    assert abs(eager_losses[0] - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss)))
    for g_pad, p_ref in zip(first_grads, ref_model.parameters()):
        np.testing.assert_allclose(g_pad.cpu().numpy(), p_ref.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)

    # ---- the same four steps as replays of one captured graph.  capture() runs the step 3 + 1 times while it warms
    # up and records (weights and seed counter move): both are put back before the comparison starts.
    buf_c, sd_c = order[:b].clone(), torch.zeros(1, dtype=torch.long, device=DEV)
    opt_c = torch.optim.Adam(m_cap.parameters(), lr=0.01, capturable=True)
    init = copy.deepcopy(m_cap.state_dict())
    replay = graphs.capture(make_step(m_cap, buf_c, sd_c, opt_c), warmup=3)
    m_cap.load_state_dict(init)
    for st in opt_c.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    sd_c.zero_()
    cap_losses = []
    for i in range(4):
        buf_c.copy_(order[i * b:(i + 1) * b])
        loss, counts = replay()
        cap_losses.append(float(loss.detach()))
    np.testing.assert_allclose(cap_losses, eager_losses, rtol=1e-6)
    for pc, pe in zip(m_cap.parameters(), m_eag.parameters()):
        np.testing.assert_allclose(pc.detach().cpu().numpy(), pe.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert int(sd_c) == 4


def test_captured_step_in_training_mode_keeps_the_sampler_flags_clean():
    """Dropout masks (bytes of 0x01) and the sampler's scratch share the graph's memory pool: the flag word of every
    replay must still start from zero (it is cleared by a kernel node; small hipMemsetAsync nodes did not replay)."""
    n, b = 30000, 256
    indptr, indices = _graph(n, 14, seed=4)
    gen = torch.Generator(device=DEV).manual_seed(2)
    x_all = torch.randn(n, 32, device=DEV, generator=gen)
    y_all = torch.randint(0, 7, (n,), device=DEV, generator=gen)
    order = torch.randperm(n, device=DEV, generator=gen)
    model, _ = _models()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    step = CapturedMiniBatchStep(indptr, indices, x_all, y_all, model.forward_padded, opt, order[:b], [10, 10], seed=5)
    seen = set()
    for i in range(6):
        loss = step(order[i * b:(i + 1) * b])
        step.check()
        assert bool(torch.isfinite(loss))
        seen.add((int(step.counts[0][0]), int(step.counts[1][0])))
    assert len(seen) > 1  # different seeds and a moving RNG seed: the replays sample different frontiers
    assert int(step.seed_dev) == 3 + 6  # 3 warm-up runs (the capture pass records, it does not execute) + 6 replays


def test_captured_step_with_the_side_branch_equals_the_single_stream_step():
    """side_stream=True: the labels and the block transposes run on a second branch of the captured graph (launched from
    the forward calls); same numbers as everything on one stream, replay after replay."""
    n, b = 30000, 256
    indptr, indices = _graph(n, 14, seed=6)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x_all = torch.randn(n, 32, device=DEV, generator=gen)
    y_all = torch.randint(0, 7, (n,), device=DEV, generator=gen)
    order = torch.randperm(n, device=DEV, generator=gen)
    m_a, m_b = _models()
    m_a.eval(), m_b.eval()  # (dropout off: number for number)
    runs = []
    for model, side in ((m_a, True), (m_b, False)):
        opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        step = CapturedMiniBatchStep(indptr, indices, x_all, y_all, model.forward_padded, opt, order[:b], [10, 10], seed=5,
                                     side_stream=side)
        losses = []
        for i in range(5):
            losses.append(float(step(order[i * b:(i + 1) * b])))
            step.check()
        runs.append((losses, [p.detach().clone() for p in model.parameters()], step.counts_table.clone()))
    assert runs[0][0] == runs[1][0]
    for pa, pb in zip(runs[0][1], runs[1][1]):
        assert torch.equal(pa, pb)
    assert torch.equal(runs[0][2], runs[1][2]) and int(runs[0][2][:, 2].sum()) == 0


REFERENCE_SCRIPT = r'''
import copy, json, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from tools import refpkg
refpkg.setup(install=True)
import torch
import torch.nn.functional as F
import cogdl_amd
from cogdl.data import Graph
from cogdl.models.nn.graphsage import Graphsage
from cogdl_amd import graphs, synth
from cogdl_amd.operators.sample import sample_adj_c
from cogdl_amd.pipeline import CapturedMiniBatchStep, HOP_SEED_STRIDE, gather_rows_by_id, sample_blocks_padded

DEV = "cuda:0"
n, b, fan = 30000, 128, [10, 10]
g = synth.scaled(n, 14, seed=3, topology="rmat", norm=None, self_loops=False)
indptr, indices = g.rowptr.long().to(DEV), g.colind.long().to(DEV)
gen = torch.Generator(device=DEV).manual_seed(1)
x_all = torch.randn(n, 32, device=DEV, generator=gen)
y_all = torch.randint(0, 7, (n,), device=DEV, generator=gen)
order = torch.randperm(n, device=DEV, generator=gen)
torch.manual_seed(0)
model = Graphsage(32, 7, [64], 2, fan, 0.5, "mean").to(DEV).eval()   # the reference's class, unchanged
twin = copy.deepcopy(model)

def adjs_of(blocks):   # what NeighborSampler hands the model (cogdl/data/sampler.py:105-116): (n_id, Graph, size)
    # (unit weights given explicitly, one per edge SLOT: left to itself Adjacency.get_weight sizes them by
    #  row_ptr[-1], cogdl/data/data.py:158-159,330-331 -- a host read, and fewer than the block has slots)
    return [(None, Graph(row_ptr=rp, col=col, edge_weight=torch.ones(col.numel(), device=DEV)), (rp.numel() - 1, n_dst))
            for (rp, col), n_dst in blocks]

def make_step(net, seeds_buf, seed_dev, opt):
    def step():
        with cogdl_amd.transient_structures():
            n_id, blocks, counts = sample_blocks_padded(indptr, indices, seeds_buf, fan, seed=77, seed_dev=seed_dev)
            opt.zero_grad(set_to_none=True)
            loss = F.cross_entropy(net(gather_rows_by_id(x_all, n_id), adjs_of(blocks)), y_all.index_select(0, seeds_buf))
            loss.backward()
            opt.step()
            seed_dev.add_(1)
        return loss, counts
    return step

# reference-shaped step on the same draws (unpadded blocks, plan cache, sizes read back on the host)
ref = copy.deepcopy(model)
seeds0, batch, blocks = order[:b].clone(), order[:b].clone(), []
for hop, k in enumerate(fan):
    rp, col, nodes, _ = sample_adj_c(indptr, indices, batch, k, False, seed=(77 + hop * HOP_SEED_STRIDE) % (1 << 64))
    blocks.append(((rp, col), batch.numel()))
    batch = nodes
ref_loss = F.cross_entropy(ref(x_all[batch], adjs_of(blocks[::-1])), y_all[seeds0])

buf_e, sd_e = order[:b].clone(), torch.zeros(1, dtype=torch.long, device=DEV)
step_e = make_step(twin, buf_e, sd_e, torch.optim.Adam(twin.parameters(), lr=0.01, capturable=True))
eager = []
for i in range(4):
    buf_e.copy_(order[i * b:(i + 1) * b])
    loss, counts = step_e()
    assert int(counts[0][2]) == 0 and int(counts[1][2]) == 0
    eager.append(float(loss.detach()))

buf_c, sd_c = order[:b].clone(), torch.zeros(1, dtype=torch.long, device=DEV)
opt_c = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
init = copy.deepcopy(model.state_dict())
replay = graphs.capture(make_step(model, buf_c, sd_c, opt_c), warmup=3)
model.load_state_dict(init)
for st in opt_c.state.values():
    for v in st.values():
        if torch.is_tensor(v):
            v.zero_()
sd_c.zero_()
captured = []
for i in range(4):
    buf_c.copy_(order[i * b:(i + 1) * b])
    loss, counts = replay()
    captured.append(float(loss.detach()))
print("REPORT " + json.dumps({"ref_loss": float(ref_loss), "eager": eager, "captured": captured}))
'''


def test_reference_graphsage_model_trains_in_a_captured_step():
    """The reference's unchanged Graphsage / SAGELayer / MeanAggregator / Graph classes (staged package) on fixed-capacity
    blocks inside cogdl_amd.transient_structures(): eager == one hipGraph replay per step == the reference-shaped step."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tools import refpkg

    if not refpkg.available():
        pytest.skip("the reference package is not staged (make -C oracle ref)")
    res = subprocess.run([sys.executable, "-c", REFERENCE_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    rep = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("REPORT ")][-1][7:])
    assert abs(rep["eager"][0] - rep["ref_loss"]) <= 1e-5 * max(1.0, abs(rep["ref_loss"]))
    np.testing.assert_allclose(rep["captured"], rep["eager"], rtol=1e-6)


def test_sampling_from_a_graph_without_nodes_flags_every_seed_and_reads_nothing():
    """Round-4 advisor: sample_prep's unconditional 16-byte load of indptr[0..1] read 8 bytes past a ONE-entry indptr
    (num_nodes == 0, batch > 0).  Every seed is invalid there: flagged, nothing loaded, an empty block comes back."""
    indptr = torch.zeros(1, dtype=torch.long, device=DEV)
    with pytest.raises(_lib.BackendError):  # (no edge array at all: refused before any launch)
        sample_adj_padded(indptr, torch.zeros(0, dtype=torch.long, device=DEV), torch.tensor([0, 5, 2], device=DEV), 3)
    indices = torch.zeros(1, dtype=torch.long, device=DEV)  # a non-null edge array: the kernels run with num_nodes == 0
    row_ptr, col, nodes, edges, counts = sample_adj_padded(indptr, indices, torch.tensor([0, 5, 2], device=DEV), 3)
    torch.cuda.synchronize()
    assert int(counts[2]) & 1 and int(counts[1]) == 0  # flagged, no edges
    assert int(row_ptr[:4].abs().sum()) == 0
