#!/usr/bin/env python3
"""Does the halo exchange of the vertex-sharded SpMM really run CONCURRENTLY with the local-block SpMM?  (Round-3 verdict:
"multi-GPU overlap is asserted, not shown".)

One GPU is all a lease has, so the wire is emulated and everything else is the product path: rank 0's shard of a 2-rank
partition of a papers100M-shaped graph is built with the HIP shard kernels, `cogdl_amd.dist._ShardedSpMM` runs unchanged
on an RCCL process group (world size 1) -- its comm stream, its events, its record_stream calls -- and only
`dist.exchange_rows`, the one call that would touch xGMI, is replaced by a stand-in that occupies the COMM STREAM for as
long as the exchange is predicted to take at the stated link rate (`predict_scaling`: bytes / (153 GB/s x 0.8)) and then
delivers rows of the right shape.  HIP events on both streams give the timeline of one forward and one backward:

    comm   |--------- all-to-all (emulated wire time) ---------|
    compute    |------ local-block csr_spmm ------|             |-- halo-block csr_spmm --|

Overlap = the intersection of the two intervals over the shorter one.  What this shows: no hidden synchronisation
(allocator, default stream, event ordering) serialises the exchange behind or in front of the local block.  What it
cannot show: RCCL's own kernels competing with the SpMM for CUs and HBM -- the 8-GPU run measures that.
Usage: python tools/overlap_timeline.py [--nodes 3470000] [--feat 128]"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import dist as cdist  # noqa: E402
from cogdl_amd.dist import HipBackend, ShardedCSR, _papers_like_shard, predict_scaling, sharded_spmm  # noqa: E402

DEV = torch.device("cuda", 0)


def build_rank0_shard(shard_nodes, degree, remote_frac, halo_frac):
    """Rank 0's shard of a 2-rank partition, assembled without a second process: the HIP split (cogdl_hip_shard_count /
    _fill) gives the two blocks and the halo table; the peer is assumed to ask for as many of our rows as we ask of it."""
    from cogdl_amd.graph_build import coo2csr_index

    rowptr, cols, w = _papers_like_shard(0, 2, shard_nodes, degree, remote_frac, 0, DEV, halo_frac)
    sh = object.__new__(ShardedCSR)
    sh.group, sh.backend, sh.rank, sh.world = None, HipBackend(), 0, 2
    sh.n_local = shard_nodes
    bounds = torch.tensor([0, shard_nodes, 2 * shard_nodes], dtype=torch.long, device=DEV)
    halo_ids, cut = sh._split_hip(rowptr, cols, w, bounds, 0, shard_nodes, 2 * shard_nodes)
    sh.n_halo = int(halo_ids.numel())
    sh.nnz_local, sh.nnz_remote = int(sh.colind_loc.numel()), int(sh.colind_rem.numel())
    sh.recv_counts = [0, sh.n_halo]
    sh.send_counts = [0, sh.n_halo]
    gen = torch.Generator(device=DEV).manual_seed(1)
    sh.send_idx = torch.randint(0, shard_nodes, (sh.n_halo,), generator=gen, device=DEV)
    sel_rowptr, order = coo2csr_index(sh.send_idx, None, shard_nodes)
    sh.sel_rowptr, sh.sel_colind = sel_rowptr.int(), order.int()
    sh._t_loc = sh._t_rem = None
    sh._comm = None
    return sh


class Wire:
    """Stand-in for dist.exchange_rows: holds the CURRENT stream (the shard's comm stream, inside _exchange_overlapped)
    for `ms` milliseconds with a spin kernel, then produces the received rows with a device copy; records its events."""

    def __init__(self, ms, clock_khz):
        self.cycles = int(ms * clock_khz)
        self.log = []

    def __call__(self, send, send_counts, recv_counts, group=None, async_op=False):
        n = int(sum(recv_counts))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(self.cycles)
        recv = send.new_empty((n,) + tuple(send.shape[1:]))
        recv.copy_(send[:n] if send.shape[0] >= n else send.new_zeros((n,) + tuple(send.shape[1:])))
        e1.record()
        self.log.append((e0, e1))
        return recv, cdist._Done()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=111_059_956 // 32, help="rows of the shard (default: 1/4 of the bench's)")
    ap.add_argument("--degree", type=float, default=28.8)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--remote-frac", type=float, default=0.1)
    ap.add_argument("--halo-frac", type=float, default=0.25)
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1)
    sh = build_rank0_shard(args.nodes, args.degree, args.remote_frac, args.halo_frac)
    x = torch.randn(args.nodes, args.feat, device=DEV, requires_grad=True)
    gout = torch.randn(args.nodes, args.feat, device=DEV)
    # the local block alone -> the rate the prediction is made from
    be = sh.backend
    for _ in range(2):
        be.spmm(sh.rowptr_loc, sh.colind_loc, sh.w_loc, x.detach())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        be.spmm(sh.rowptr_loc, sh.colind_loc, sh.w_loc, x.detach())
    e1.record()
    torch.cuda.synchronize()
    loc_ms = e0.elapsed_time(e1) / 5
    pred = predict_scaling(args.nodes, sh.nnz_local + sh.nnz_remote, args.feat, sh.nnz_remote / (sh.nnz_local + sh.nnz_remote),
                           sh.n_halo, loc_ms / (sh.nnz_local / 1e9))
    wire_ms = pred["2"]["a2a_ms"]
    # spin-kernel calibration: cycles per millisecond of torch.cuda._sleep on this device
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    torch.cuda._sleep(100_000_000)
    c1.record()
    torch.cuda.synchronize()
    khz = 100_000_000 / c0.elapsed_time(c1)
    print("shard: %d rows, %d local + %d remote edges, halo %d rows = %.3f GB per direction; local block alone %.2f ms; emulated "
          "wire time %.2f ms (one xGMI link at %.0f GB/s x %.2f)" % (args.nodes, sh.nnz_local, sh.nnz_remote, sh.n_halo,
                                                                        sh.n_halo * args.feat * 4 / 1e9, loc_ms, wire_ms,
                                                                        cdist.XGMI_LINK_GBS, cdist.XGMI_LINK_EFF), flush=True)
    # instrument the local-block launches: events around every backend.spmm call, tagged by the structure it runs on
    spans = []
    real_spmm = HipBackend.spmm

    def traced(self, rowptr, colind, val, xx, out=None):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = real_spmm(self, rowptr, colind, val, xx, out=out)
        b.record()
        spans.append((a, b, int(colind.numel()), out is not None))
        return r

    HipBackend.spmm = traced
    for wire_scale, label in ((1.0, "predicted wire time"), (3.0, "3 x the predicted wire time (exchange longer than the local block)")):
        wire = Wire(wire_ms * wire_scale, khz)
        cdist.exchange_rows = wire
        sh.transposed()
        for it in range(4):  # (the first passes warm the allocator and the clocks; the last one is reported)
            spans.clear()
            wire.log.clear()
            torch.cuda.synchronize()
            t_ref = torch.cuda.Event(enable_timing=True)
            t_ref.record()
            t0 = time.perf_counter()
            y = sharded_spmm(sh, x)
            x.grad = None
            y.backward(gout)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
        print("\n== %s: one forward + backward, %.2f ms wall" % (label, wall))
        rows = []
        for (a, b), name in zip(wire.log, ("forward  all-to-all (comm stream)", "backward all-to-all (comm stream)")):
            rows.append((t_ref.elapsed_time(a), t_ref.elapsed_time(b), name))
        names = iter(["forward  local-block csr_spmm", "forward  halo-block csr_spmm_acc", "backward halo-block transpose csr_spmm",
                      "backward local-block transpose csr_spmm", "backward accumulate returned rows (csr_spmm_acc)"])
        for a, b, nnz, acc in spans:
            rows.append((t_ref.elapsed_time(a), t_ref.elapsed_time(b), next(names, "csr_spmm") + " [%d edges]" % nnz))
        for s0, s1, name in sorted(rows):
            print("  %8.3f -> %8.3f ms  (%7.3f ms)  %s" % (s0, s1, s1 - s0, name))
        comm = [r for r in rows if "all-to-all" in r[2]]
        local = [r for r in rows if "local-block" in r[2]]
        for c, l in zip(comm, local):
            inter = max(0.0, min(c[1], l[1]) - max(c[0], l[0]))
            shorter = min(c[1] - c[0], l[1] - l[0])
            print("  overlap of [%s] with [%s]: %.3f ms = %.0f %% of the shorter interval" % (
                c[2].split(" (")[0], l[2].split(" [")[0], inter, 100 * inter / max(shorter, 1e-9)))
        serial = sum(r[1] - r[0] for r in rows)
        print("  sum of all intervals %.2f ms vs %.2f ms from the first start to the last end: %.2f ms hidden by the overlap" % (
            serial, max(r[1] for r in rows) - min(r[0] for r in rows), serial - (max(r[1] for r in rows) - min(r[0] for r in rows))))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
