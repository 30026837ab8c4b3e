"""The vertex-sharded SpMM (cogdl_amd/dist.py) with world_size 2 and 3 on CPU: gloo processes, local kernels
injected from the oracle (the product default is the HIP backend).  Checks sharded == unsharded for the forward
and for the gradient, the halo bookkeeping, and degenerate shards."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    """CPU stand-in for the HIP kernels (tests only)."""

    def spmm(self, rowptr, colind, val, x, out=None):
        from oracle import oracle

        y = torch.from_numpy(oracle.csr_spmm(rowptr, colind, val, x.detach()))
        return y if out is None else out.add_(y)

    def transpose(self, rowptr, colind, val, n_cols):
        from oracle import oracle

        colptr, rowind, val_t, _ = oracle.csr2csc(rowptr, colind, val, n_cols=n_cols)
        return torch.from_numpy(colptr), torch.from_numpy(rowind), None if val_t is None else torch.from_numpy(val_t)


def _worker(rank, world, port, n, seed, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cogdl_amd import synth
        from cogdl_amd.dist import ShardedCSR, partition_bounds, sharded_spmm

        g = synth.scaled(n, 6, seed=seed)  # every rank builds the same global graph, keeps its rows
        bounds = partition_bounds(n, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        e0, e1 = int(g.rowptr[lo]), int(g.rowptr[hi])
        rowptr = (g.rowptr[lo:hi + 1] - g.rowptr[lo]).long()
        sh = ShardedCSR(rowptr, g.colind[e0:e1].long(), g.weight[e0:e1], bounds, backend=OracleBackend())
        assert sh.nnz_local + sh.nnz_remote == e1 - e0
        assert sum(sh.recv_counts) == sh.n_halo and sh.recv_counts[rank] == 0
        x = torch.randn(n, 12, generator=torch.Generator().manual_seed(seed))
        gout = torch.randn(n, 12, generator=torch.Generator().manual_seed(seed + 1))
        xl = x[lo:hi].clone().requires_grad_()
        y = sharded_spmm(sh, xl)
        y.backward(gout[lo:hi])
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), y=y.detach().numpy(), gx=xl.grad.numpy(), lo=lo, hi=hi,
                 n_halo=sh.n_halo, send=sum(sh.send_counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 300), (3, 301), (2, 5)])
def test_sharded_equals_unsharded(tmp_path, oracle, world, n):
    from cogdl_amd import synth

    port = 29600 + world * 10 + (n % 7)
    mp.spawn(_worker, args=(world, port, n, 3, str(tmp_path)), nprocs=world, join=True)
    g = synth.scaled(n, 6, seed=3)
    x = torch.randn(n, 12, generator=torch.Generator().manual_seed(3))
    gout = torch.randn(n, 12, generator=torch.Generator().manual_seed(4))
    want_y = oracle.csr_spmm_f64(g.rowptr, g.colind, g.weight, x)
    colptr, rowind, w_t, _ = oracle.csr2csc(g.rowptr, g.colind, g.weight, n_cols=n)
    want_gx = oracle.csr_spmm_f64(colptr, rowind, w_t, gout)
    total_halo = total_send = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        lo, hi = int(z["lo"]), int(z["hi"])
        np.testing.assert_allclose(z["y"], want_y[lo:hi], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(z["gx"], want_gx[lo:hi], rtol=1e-5, atol=1e-6)
        total_halo += int(z["n_halo"])
        total_send += int(z["send"])
    assert total_halo == total_send  # every requested halo row is sent by exactly one owner


def test_papers_like_shard_generator_shape():
    """bench.py's N>1 workload: remote sources come from per-peer boundary slices, so the halo is bounded."""
    from cogdl_amd.dist import _papers_like_shard

    world, s, deg = 4, 20000, 12.0
    for rank in range(world):
        rowptr, cols, w = _papers_like_shard(rank, world, s, deg, 0.1, 0, "cpu", halo_frac=0.25)
        assert rowptr.numel() == s + 1 and int(rowptr[-1]) == cols.numel() == w.numel()
        assert int(cols.min()) >= 0 and int(cols.max()) < world * s
        owner = cols // s
        remote = owner != rank
        assert 0.07 < float(remote.float().mean()) < 0.12  # 10 % of the random edges (self loops are local)
        halo = torch.unique(cols[remote])
        pool = int(0.25 * s / (world - 1))
        assert halo.numel() <= (world - 1) * pool
        for q in range(world):  # this rank's slice of every peer's boundary region
            if q != rank:
                off = halo[(halo // s) == q] - q * s
                slot = (rank - q - 1) % world
                assert int(off.min()) >= slot * pool and int(off.max()) < (slot + 1) * pool
        sums = torch.zeros(s).index_add_(0, torch.repeat_interleave(torch.arange(s), rowptr[1:] - rowptr[:-1]), w)
        assert torch.allclose(sums, torch.ones(s), atol=1e-5)
    # worst case: uniform remote sources -> about one halo row per remote edge
    rowptr, cols, _ = _papers_like_shard(0, world, s, deg, 0.1, 0, "cpu", halo_frac=0.0)
    remote = (cols // s) != 0
    assert torch.unique(cols[remote]).numel() > 0.6 * int(remote.sum())
