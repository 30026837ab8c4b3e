"""The vertex-sharded SpMM on the real backend: RCCL process group (world size 1 -- the GPU box has one device, and
RCCL refuses two ranks on one GPU) + the HIP kernels (csr_spmm, csr_spmm_acc, csr2csc, gather).  With a single rank
the halo is empty, so a second check splits one graph into two column blocks by hand and drives the same
local-block / remote-block code path (Y = A_loc X_loc; Y += A_rem X_halo) the multi-rank forward uses."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from cogdl_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def rccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def test_sharded_world1_matches_unsharded(rccl_world1, oracle):
    from cogdl_amd.dist import ShardedCSR, partition_bounds, sharded_spmm

    g = synth.scaled(5000, 10, seed=1)
    bounds = partition_bounds(g.num_nodes, 1)
    sh = ShardedCSR(g.rowptr.to(DEV).long(), g.colind.to(DEV).long(), g.weight.to(DEV), bounds)
    assert sh.n_halo == 0 and sh.nnz_local == g.nnz
    x = torch.randn(g.num_nodes, 64, generator=torch.Generator().manual_seed(2))
    gout = torch.randn(g.num_nodes, 64, generator=torch.Generator().manual_seed(3))
    xd = x.to(DEV).requires_grad_()
    y = sharded_spmm(sh, xd)
    y.backward(gout.to(DEV))
    want = oracle.csr_spmm(g.rowptr, g.colind, g.weight, x)
    assert y.detach().cpu().numpy().tobytes() == want.tobytes()
    colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=g.num_nodes)
    want_g = oracle.csr_spmm(colptr, rowind, w_t, gout)
    assert xd.grad.cpu().numpy().tobytes() == want_g.tobytes()


def test_all_to_all_uneven_rows_world1(rccl_world1):
    from cogdl_amd.dist import exchange_rows

    send = torch.arange(12, dtype=torch.float32, device=DEV).view(6, 2)
    recv, work = exchange_rows(send, [6], [6], async_op=True)
    work.wait()
    assert torch.equal(recv, send)


def test_local_plus_halo_blocks_equal_whole(oracle):
    """Rank 0 of a 2-way partition, emulated: the halo rows are fetched by index instead of by all-to-all."""
    from cogdl_amd.dist import HipBackend

    g = synth.scaled(4000, 12, seed=5)
    n, half = g.num_nodes, g.num_nodes // 2
    rowptr, colind, w = g.rowptr[:half + 1].long(), g.colind[:int(g.rowptr[half])].long(), g.weight[:int(g.rowptr[half])]
    rows = torch.repeat_interleave(torch.arange(half), (rowptr[1:] - rowptr[:-1]))
    loc = colind < half

    def csr(mask, cols):
        cnt = torch.bincount(rows[mask], minlength=half)
        rp = torch.zeros(half + 1, dtype=torch.long)
        rp[1:] = torch.cumsum(cnt, 0)
        return rp.int().to(DEV), cols.int().to(DEV), w[mask].to(DEV)

    halo_ids, inv = torch.unique(colind[~loc], return_inverse=True)
    rp_l, ci_l, w_l = csr(loc, colind[loc])
    rp_r, ci_r, w_r = csr(~loc, inv)
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(6))
    be = HipBackend()
    y = be.spmm(rp_l, ci_l, w_l, x[:half].to(DEV))
    y = be.spmm(rp_r, ci_r, w_r, x[halo_ids].to(DEV), out=y)
    want = oracle.csr_spmm_f64(g.rowptr, g.colind, g.weight, x)[:half]
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def _gpu_worker(rank, world, port, n, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU: gloo, not RCCL
    try:
        from cogdl_amd import synth as sy
        from cogdl_amd.dist import ShardedCSR, partition_bounds, sharded_spmm

        torch.cuda.set_device(0)
        g = sy.scaled(n, 8, seed=11, topology="rmat")  # same global graph on every rank; hub rows included
        bounds = partition_bounds(n, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = int(g.rowptr[lo]), int(g.rowptr[hi])
        rowptr = (g.rowptr[lo:hi + 1] - g.rowptr[lo]).long().to(DEV)
        sh = ShardedCSR(rowptr, g.colind[e0:e1].long().to(DEV), g.weight[e0:e1].to(DEV), bounds)  # HipBackend
        x = torch.randn(n, 32, generator=torch.Generator().manual_seed(5))
        gout = torch.randn(n, 32, generator=torch.Generator().manual_seed(6))
        xl = x[lo:hi].to(DEV).requires_grad_()
        y = sharded_spmm(sh, xl)
        y.backward(gout[lo:hi].to(DEV))
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), y=y.detach().cpu().numpy(), gx=xl.grad.cpu().numpy(), lo=lo,
                 hi=hi, n_halo=sh.n_halo)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_hip_kernels_with_halo(tmp_path, oracle):
    """World size 2 with a real halo: both ranks run the HIP kernels (local block, remote block via csr_spmm_acc,
    transposed blocks in backward) on the one GPU, rows travel through gloo.  sharded == unsharded."""
    import torch.multiprocessing as mp

    n, world = 6000, 2
    mp.spawn(_gpu_worker, args=(world, 29641, n, str(tmp_path)), nprocs=world, join=True)
    g = synth.scaled(n, 8, seed=11, topology="rmat")
    x = torch.randn(n, 32, generator=torch.Generator().manual_seed(5))
    gout = torch.randn(n, 32, generator=torch.Generator().manual_seed(6))
    want_y = oracle.csr_spmm_f64(g.rowptr, g.colind, g.weight, x)
    colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=n)
    want_gx = oracle.csr_spmm_f64(colptr, rowind, w_t, gout)
    halo = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        lo, hi = int(z["lo"]), int(z["hi"])
        np.testing.assert_allclose(z["y"], want_y[lo:hi], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(z["gx"], want_gx[lo:hi], rtol=1e-5, atol=1e-5)
        halo += int(z["n_halo"])
    assert halo > 0


def test_captured_minibatch_step_with_the_gradient_all_reduce_inside_the_graph(rccl_world1):
    """configs[3] replicas: CapturedMiniBatchStep(process_group=...) puts one flat RCCL all-reduce of the gradients
    between backward and the optimizer INSIDE the captured hipGraph.  With the one rank this box has the average is
    the identity: the replays must equal those of the step captured without a group, number for number."""
    import copy

    from cogdl_amd.pipeline import CapturedMiniBatchStep
    from tools.sage_bench import Sage

    n, b = 20000, 128
    g = synth.scaled(n, 12, seed=6, topology="rmat", norm=None, self_loops=False)
    indptr, indices = g.rowptr.long().to(DEV), g.colind.long().to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x_all = torch.randn(n, 32, device=DEV, generator=gen)
    y_all = torch.randint(0, 7, (n,), device=DEV, generator=gen)
    order = torch.randperm(n, device=DEV, generator=gen)
    torch.manual_seed(0)
    m_a = Sage(32, 64, 7).to(DEV).eval()
    m_b = copy.deepcopy(m_a)
    losses = []
    for model, group in ((m_a, None), (m_b, dist.group.WORLD)):
        opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        step = CapturedMiniBatchStep(indptr, indices, x_all, y_all, model.forward_padded, opt, order[:b], [10, 10], seed=1,
                                     process_group=group)
        out = []
        for i in range(4):
            out.append(float(step(order[(i + 1) * b:(i + 2) * b])))
        step.check()
        losses.append(out)
    np.testing.assert_allclose(losses[1], losses[0], rtol=1e-6)
    for pa, pb in zip(m_a.parameters(), m_b.parameters()):
        np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_bench_gpus_2_orchestration_on_the_one_gpu():
    """`python bench.py --gpus 2 --share-gpu`: the whole N > 1 bench orchestration on this one-GPU box -- bench.py spawns
    its two ranks itself; the MAIN leg shards the papers100M-shaped graph itself (1/64 scale in this mode: SURVEY section
    8d's size) by edge-balanced row ranges, both ranks build their shard with the HIP kernels (measured halos), run the
    sharded step; then the generated-shard leg (`assumed_partition`) and the worst-case-partition leg run in child
    interpreters with a process group of their own; rows travel through gloo because RCCL refuses two ranks on one device.
    The line must report two ranks in every leg, and the halo it reports must equal dist.halo_rows on the same partition."""
    import json
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--shard-nodes",
                           "200000", "--steps", "2", "--warmup", "1", "--feat", "64"], capture_output=True, text=True,
                          timeout=900, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, proc.stderr[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and len(line["local_block_ms_by_rank"]) == 2
    assert line["scaling"] == "strong" and line["config"]["scale"] == 64 and "predicted" not in line
    assert line["config"]["halo_rows_rank0"] > 0 and line["halo_GB_per_step_all_ranks"] > 0
    # the reported halo / remote edges are what dist.halo_rows counts on the same graph and bounds
    from cogdl_amd import synth
    from cogdl_amd.dist import edge_balanced_bounds, halo_rows

    g = synth.papers100m_like("cuda:0", True, 0, synth.PAPERS_NODES // 64, synth.PAPERS_PAIRS // 64)
    bounds = edge_balanced_bounds(g.rowptr, 2)
    assert line["config"]["bounds"] == bounds.tolist() and line["config"]["nnz_global"] == g.nnz
    counted = halo_rows(g.rowptr, g.colind, bounds)
    assert [c[0] for c in counted] == line["config"]["remote_edges_by_rank"]
    assert [c[1] for c in counted] == line["config"]["halo_rows_by_rank"]
    del g
    torch.cuda.empty_cache()
    assumed = line["assumed_partition"]
    assert assumed.get("n_gpus") == 2 and assumed.get("n_ranks_seen") == 2 and assumed["scaling"] == "weak", assumed
    assert assumed["config"]["nodes_per_gpu"] == 200000 and "predicted" in assumed
    worst = line["worst_case_partition"]
    assert worst.get("n_gpus") == 2 and worst.get("n_ranks_seen") == 2 and worst["config"]["remote_frac"] == 0.5, worst
