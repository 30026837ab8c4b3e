"""Vertex-sharded csr_spmm for graphs larger than one GPU (BASELINE.json configs[4]: GCN on
ogbn-papers100M over 8 x MI355X).  No reference counterpart: CogDL only has DDP replicas
(cogdl/trainer/trainer.py:291-303); the single-GPU operator semantics (operators/spmm.py:43-80) are
kept -- sharded(A, X) equals csr_spmm(A, X) on the unsharded graph.

Layout (1-D row partition, one process per GPU, torch.distributed over RCCL/xGMI):
    rank p owns rows [bounds[p], bounds[p+1]) of A (all their incoming edges) and the same rows of X, Y.
    A_p is split by column ownership into
        A_loc  : columns owned by p, relabelled to local row ids           -> int32 CSR
        A_rem  : all other columns, relabelled into a compact HALO table   -> int32 CSR
    halo = sorted unique remote column ids, grouped by owner (owners are contiguous id ranges, so
    "sorted by id" is "sorted by owner"); int32 indices stay valid at papers100M scale because both
    tables are shard-local (SURVEY.md section 7, int32 limits).
Forward per call:
    1. gather the rows other ranks asked for            cogdl_hip_gather_feature_rows     (compute stream)
    2. all-to-all the halo rows                         RCCL, on an explicit COMM stream behind an event
    3. Y  = A_loc . X_local                             overlaps with 2                   (compute stream)
    4. Y += A_rem . halo                                after the comm stream's event (cogdl_hip_csr_spmm_acc)
Backward is the transposed pattern: G_halo = A_rem^T . G (sent back with the reverse all-to-all,
overlapped with A_loc^T . G) and accumulated into the owners' rows by ONE kernel: the returned rows are
summed per owner row through a selection matrix S (S[r, j] = 1 iff returned row j belongs to local row r;
built once per shard, its entries in peer order), i.e. G_x += S . back with cogdl_hip_csr_spmm_acc --
deterministic (fixed order per row), no atomics, no per-peer launches.
xGMI is point-to-point: the all-to-all drives all 7 links of a GPU at once (one send/recv pair per
peer), which is why the exchange is an all-to-all and not a ring all-gather of X.

The local SpMM kernels are injected (`backend`) so that the partitioning / exchange logic is testable
on CPU with gloo; the default backend is the HIP one and raises if the library is missing.
"""
import os
import time

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------- backends
class HipBackend:
    """Local kernels on the GPU through libcogdl_hip (the product path)."""

    def spmm(self, rowptr, colind, val, x, out=None):
        from .operators.spmm import csr_spmm_raw

        return csr_spmm_raw(rowptr, colind, val, x, out=out)

    def gather(self, x, idx):
        from .pipeline import gather_rows_by_id

        return gather_rows_by_id(x.detach(), idx)

    def transpose(self, rowptr, colind, val, n_cols):
        from .plan import csr2csc, gather_rows

        plan = csr2csc(rowptr, colind, n_cols)
        return plan.colptr, plan.rowind, (gather_rows(plan.perm, val) if val is not None else None)


# ------------------------------------------------------------------------------------- exchange
class _Done:
    def wait(self):
        pass


def exchange_rows(send, send_counts, recv_counts, group=None, async_op=False):
    """Variable-size row exchange: `send` holds, back to back, send_counts[q] rows for every rank q;
    returns (recv, work) with recv_counts[q] rows from every rank q, in rank order."""
    world = dist.get_world_size(group)
    recv = send.new_empty((int(sum(recv_counts)),) + tuple(send.shape[1:]))
    if dist.get_backend(group) == "nccl":  # RCCL on ROCm
        work = dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts),
                                      input_split_sizes=list(send_counts), group=group, async_op=async_op)
        return recv, (work if async_op else _Done())
    # gloo (tests) has no all_to_all: pairwise isend/irecv; GPU tensors are staged through the host (gloo moves host
    # memory only) -- this is how the world-size-2 test drives the HIP kernels on a box with a single GPU
    if send.is_cuda:
        recv_host, _ = exchange_rows(send.cpu(), send_counts, recv_counts, group)
        recv.copy_(recv_host)
        return recv, _Done()
    rank = dist.get_rank(group)
    s_off = [0]
    r_off = [0]
    for q in range(world):
        s_off.append(s_off[-1] + int(send_counts[q]))
        r_off.append(r_off[-1] + int(recv_counts[q]))
    recv[r_off[rank]:r_off[rank + 1]] = send[s_off[rank]:s_off[rank + 1]]
    ops = []
    for q in range(world):
        if q == rank:
            continue
        if send_counts[q]:
            ops.append(dist.P2POp(dist.isend, send[s_off[q]:s_off[q + 1]].contiguous(), q, group))
        if recv_counts[q]:
            ops.append(dist.P2POp(dist.irecv, recv[r_off[q]:r_off[q + 1]], q, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv, _Done()


def partition_bounds(num_nodes, world):
    """Contiguous, near-equal row ranges: bounds[p] .. bounds[p+1]."""
    base, rem = divmod(int(num_nodes), world)
    b = [0]
    for p in range(world):
        b.append(b[-1] + base + (1 if p < rem else 0))
    return torch.tensor(b, dtype=torch.long)


def _csr_from_sorted_rows(rows, n_rows):
    counts = torch.bincount(rows, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.long, device=rows.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr


# ------------------------------------------------------------------------------------ the shard
class ShardedCSR:
    """One rank's shard of a row-partitioned CSR matrix plus its halo exchange plan."""

    def __init__(self, rowptr, colind_global, weight, bounds, group=None, backend=None):
        """rowptr [n_local+1] (any int dtype), colind_global [nnz] GLOBAL column ids (int64),
        weight [nnz] fp32 or None, bounds [world+1] (partition_bounds)."""
        self.group = group
        self.backend = backend or HipBackend()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = colind_global.device
        bounds = bounds.to(dev)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        self.n_local = hi - lo
        assert rowptr.numel() == self.n_local + 1, "rowptr does not match this rank's row range"
        deg = (rowptr[1:] - rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(self.n_local, device=dev), deg)
        col = colind_global.long()
        is_local = (col >= lo) & (col < hi)

        def sub(mask):
            return rows[mask], col[mask], (weight[mask] if weight is not None else None)

        r_l, c_l, self.w_loc = sub(is_local)
        r_r, c_r, self.w_rem = sub(~is_local)
        self.rowptr_loc = _csr_from_sorted_rows(r_l, self.n_local).int()
        self.colind_loc = (c_l - lo).int()
        halo_ids, inv = torch.unique(c_r, return_inverse=True)  # sorted => grouped by owner
        self.rowptr_rem = _csr_from_sorted_rows(r_r, self.n_local).int()
        self.colind_rem = inv.int()
        self.n_halo = int(halo_ids.numel())
        self.nnz_local, self.nnz_remote = int(c_l.numel()), int(c_r.numel())
        # how many halo rows come from each owner, and which of MY rows each peer wants
        cut = torch.searchsorted(halo_ids, bounds)
        self.recv_counts = [int(cut[q + 1] - cut[q]) for q in range(self.world)]
        counts_t = torch.tensor(self.recv_counts, dtype=torch.long, device=dev)
        want_t, _ = exchange_rows(counts_t.view(-1, 1), [1] * self.world, [1] * self.world, group)
        self.send_counts = [int(v) for v in want_t.view(-1).tolist()]
        send_ids, _ = exchange_rows(halo_ids, self.recv_counts, self.send_counts, group)
        self.send_idx = (send_ids - lo).long()  # local row ids, grouped by requesting rank
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.n_local)
        # Selection matrix of the backward accumulation: row r lists the positions j of `back` (= of send_idx) that
        # belong to local row r, in peer order (stable sort) -> gx += S . back is one deterministic csr_spmm_acc.
        order = torch.sort(self.send_idx, stable=True).indices
        self.sel_rowptr = _csr_from_sorted_rows(self.send_idx[order], self.n_local).int()
        self.sel_colind = order.int()
        self._t_loc = self._t_rem = None
        self._comm = None  # explicit communication stream (GPU shards), created on first use

    # transposes for the backward pass, built on first use
    def transposed(self):
        if self._t_loc is None:
            self._t_loc = self.backend.transpose(self.rowptr_loc, self.colind_loc, self.w_loc, self.n_local)
            self._t_rem = self.backend.transpose(self.rowptr_rem, self.colind_rem, self.w_rem, self.n_halo)
        return self._t_loc, self._t_rem

    def halo_bytes(self, feat, elem=4):
        return self.n_halo * feat * elem


def _exchange_overlapped(sh, send, send_counts, recv_counts):
    """The halo all-to-all on the shard's COMM stream, ordered behind everything enqueued on the compute stream so
    far (event) -- returns (recv, done) where done() makes the compute stream wait for the exchange.  The local-block
    SpMM the caller enqueues in between overlaps with it.  CPU / gloo shards (tests) exchange synchronously."""
    if not send.is_cuda or dist.get_backend(sh.group) != "nccl":
        recv, work = exchange_rows(send, send_counts, recv_counts, sh.group, async_op=True)
        return recv, work.wait
    if sh._comm is None:
        sh._comm = torch.cuda.Stream(device=send.device)
    compute = torch.cuda.current_stream(send.device)
    ready = torch.cuda.Event()
    ready.record(compute)
    with torch.cuda.stream(sh._comm):
        sh._comm.wait_event(ready)
        recv, _ = exchange_rows(send, send_counts, recv_counts, sh.group, async_op=False)
        finished = torch.cuda.Event()
        finished.record(sh._comm)
    send.record_stream(sh._comm)
    recv.record_stream(compute)
    return recv, lambda: compute.wait_event(finished)


class _ShardedSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh):
        be = sh.backend
        send = be.gather(x, sh.send_idx) if hasattr(be, "gather") else x.index_select(0, sh.send_idx)
        halo, done = _exchange_overlapped(sh, send, sh.send_counts, sh.recv_counts)
        y = be.spmm(sh.rowptr_loc, sh.colind_loc, sh.w_loc, x)  # overlaps with the all-to-all
        done()
        if sh.n_halo:
            y = be.spmm(sh.rowptr_rem, sh.colind_rem, sh.w_rem, halo, out=y)
        ctx.sh = sh
        return y

    @staticmethod
    def backward(ctx, g):
        sh = ctx.sh
        be = sh.backend
        g = g.contiguous()
        (cp_l, ri_l, w_l), (cp_r, ri_r, w_r) = sh.transposed()
        g_halo = be.spmm(cp_r, ri_r, w_r, g) if sh.n_halo else g.new_zeros((0, g.shape[1]))
        back, done = _exchange_overlapped(sh, g_halo, sh.recv_counts, sh.send_counts)
        gx = be.spmm(cp_l, ri_l, w_l, g)  # overlaps with the reverse all-to-all
        done()
        if back.shape[0]:
            gx = be.spmm(sh.sel_rowptr, sh.sel_colind, None, back, out=gx)  # gx += S . back: fixed order, one launch
        return gx, None


def sharded_spmm(sh, x_local):
    """Y_local = (A X)[my rows]; differentiable w.r.t. x_local."""
    return _ShardedSpMM.apply(x_local.contiguous(), sh)


# ------------------------------------------------------------------------------- bench (N > 1)
def _papers_like_shard(rank, world, shard_nodes, degree, remote_frac, seed, device, halo_frac=0.25):
    """This rank's rows of a papers100M-shaped graph: `shard_nodes` rows, mean in-degree `degree`; a fraction
    `remote_frac` of every row's sources lies in OTHER shards (owner uniform over them), the rest inside the own shard
    -- the shape a locality-preserving (METIS-like) 1-D partition of a citation graph has.  The remote sources of the
    pair (this rank, owner q) come from a BOUNDARY region of q: a contiguous slice of halo_frac * shard_nodes /
    (world - 1) nodes, one slice per requesting rank, so that a rank's halo table holds about halo_frac * shard_nodes
    rows (ghost-node ratios of 0.2-0.4 are what METIS partitions of citation graphs show).  halo_frac <= 0: remote
    sources uniform over the whole owner shard (no reuse: the halo then approaches one row per remote EDGE);
    together with remote_frac = (world-1)/world that is a random partition of a structureless graph, the worst case
    for a 1-D partition.  Self loop appended, row-normalised weights."""
    gdev = torch.device(device)
    g = torch.Generator(device=gdev).manual_seed(seed * 1000003 + rank)  # generated where it is used: a true
    nnz = int(shard_nodes * degree)                                       # papers100M shard is 4e8 edges per GPU
    lo = rank * shard_nodes

    def randint(high):
        return torch.randint(0, high, (nnz,), generator=g, device=gdev)

    rows = randint(shard_nodes)
    cols = randint(shard_nodes)  # offset inside a shard
    if world > 1:
        is_remote = torch.rand(nnz, generator=g, device=gdev) < remote_frac
        other = randint(world - 1)
        other = other + (other >= rank).long()  # uniform over the other shards
        owner = torch.where(is_remote, other, torch.full_like(other, rank))
        del is_remote, other
    else:
        owner = torch.zeros(nnz, dtype=torch.long, device=gdev)
    if world > 1 and halo_frac > 0:
        pool = max(1, int(halo_frac * shard_nodes / (world - 1)))
        slot = (rank - owner - 1) % world  # 0 .. world-2 for owner != rank: this rank's slice of the owner's boundary
        boundary = slot * pool + randint(pool)
        cols = torch.where(owner != rank, boundary, cols)
        del slot, boundary
    cols = cols + owner * shard_nodes
    del owner
    rows = torch.cat([rows, torch.arange(shard_nodes, device=device)])
    cols = torch.cat([cols, torch.arange(lo, lo + shard_nodes, device=device)])
    order = torch.sort(rows, stable=True).indices
    rows, cols = rows[order], cols[order]
    rowptr = _csr_from_sorted_rows(rows, shard_nodes)
    deg = (rowptr[1:] - rowptr[:-1]).float()
    w = (1.0 / deg)[rows]
    return rowptr, cols, w


def _rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def bench_sharded_spmm(args):
    """Weak-scaling bench of the vertex-sharded csr_spmm forward + backward (bench.py --gpus N, N > 1)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # Weak scaling with the TRUE papers100M shard per GPU: 111,059,956 nodes / 8 = 13.9 M rows, ~4.1e8 edges, X = 7.1 GB
    # per GPU -- at N = 8 this is the whole graph (3.3e9 edges: beyond what int32 CSR indices can address on ONE GPU,
    # which is why the single-GPU leg of the curve cannot be the unsharded graph and the scaling is weak, not strong).
    shard_nodes = args.shard_nodes or 111_059_956 // 8
    degree = args.shard_degree or 28.8                         # 3.2e9 symmetrised edges / 111e6 nodes
    f = args.feat
    remote_frac = args.remote_frac if args.remote_frac >= 0 else 0.1
    halo_frac = getattr(args, "halo_frac", 0.25)
    rowptr, cols, w = _papers_like_shard(rank, world, shard_nodes, degree, remote_frac, 0, dev, halo_frac)
    bounds = torch.arange(world + 1, dtype=torch.long) * shard_nodes
    sh = ShardedCSR(rowptr, cols, w, bounds)
    del cols
    x = torch.randn(shard_nodes, f, device=dev, requires_grad=True)
    gout = torch.randn(shard_nodes, f, device=dev)
    nnz_global = torch.tensor([sh.nnz_local + sh.nnz_remote], device=dev, dtype=torch.long)
    dist.all_reduce(nnz_global)
    nnz_global = int(nnz_global)

    def step():
        y = sharded_spmm(sh, x)
        x.grad = None
        y.backward(gout)

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)

    # the local-only part alone (no exchange): what one GPU does on the same shard, for reference
    from .operators.spmm import csr_spmm_raw

    with torch.no_grad():
        for _ in range(3):
            csr_spmm_raw(sh.rowptr_loc, sh.colind_loc, sh.w_loc, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            csr_spmm_raw(sh.rowptr_loc, sh.colind_loc, sh.w_loc, x)
        e1.record()
        torch.cuda.synchronize()
        loc_ms = e0.elapsed_time(e1) / 10
    halo_gb = torch.tensor([sh.halo_bytes(f)], device=dev, dtype=torch.float64)
    dist.all_reduce(halo_gb)
    result = None
    if rank == 0:
        result = {
            "metric": "SpMM GEdges/s (vertex-sharded csr_spmm fwd+bwd, papers100M-shaped shards) @%d GPUs" % world,
            "value": 2 * nnz_global * args.steps / dt / 1e9,
            "unit": "GEdges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "papers100M-like 1-D vertex-sharded csr_spmm fwd+bwd (configs[4]); %.0f%% of every "
                                   "row's sources in other shards, drawn from boundary regions sized for a halo of "
                                   "%.2f x the shard's rows (locality-preserving partition; --remote-frac %.3f "
                                   "--halo-frac 0 = random partition, worst-case halo)"
                                   % (100 * remote_frac, max(halo_frac, 0.0), (world - 1) / world),
                       "nodes_per_gpu": shard_nodes, "nnz_global": nnz_global, "feat": f, "remote_frac": remote_frac,
                       "halo_frac": halo_frac, "halo_rows_rank0": sh.n_halo,
                       "parallelism": "vertex-shard x%d, RCCL all-to-all halo exchange overlapped with local SpMM" % world},
            "halo_GB_per_step_all_ranks": float(halo_gb) * 2 / 1e9,
            "n_ranks_seen": dist.get_world_size(), "rccl_version": _rccl_version(),
            "exchange": "all_to_all_single on an explicit comm stream (event-ordered), gather = cogdl_hip_gather_feature_rows, "
                        "backward accumulation = one csr_spmm_acc over the selection matrix",
            "local_block_spmm_ms_rank0": loc_ms,
            "local_block_GEdges_s_rank0": sh.nnz_local / (loc_ms * 1e-3) / 1e9,
        }
        # the dominant kernel of a step: csr_spmm over the rank's local block (SURVEY.md section 8d's formula), timed
        # above with HIP events on its own; X (7.1 GB per shard) is far beyond the caches, so this IS HBM traffic
        b_alg = sh.nnz_local * (4 + 4 + f * 4) + shard_nodes * (4 + f * 4)
        result["roofline"] = {"bound": "hbm", "kernel": "rowreduce_main_kernel<SpmmOp<float,...>> on the local block A_pp (rank 0)",
                              "achieved": b_alg / (loc_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": b_alg / (loc_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                              "algorithmic_bytes_per_launch": b_alg}
    dist.barrier()
    dist.destroy_process_group()
    return result
