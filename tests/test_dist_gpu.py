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
