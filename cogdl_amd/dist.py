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

_T_IMPORT = time.time()  # bench legs are budgeted against the time since this process started working


# ------------------------------------------------------------------------------------- backends
class HipBackend:
    """Local kernels on the GPU through libcogdl_hip (the product path)."""

    def spmm(self, rowptr, colind, val, x, out=None, order=None):
        from .operators.spmm import csr_spmm_raw

        return csr_spmm_raw(rowptr, colind, val, x, out=out, row_order=order)

    def row_schedule(self, rowptr, colind):
        """The row blocks' schedule of a shard's block (cogdl_amd/bigcsr.py: rows by decreasing degree inside windows; papers100M-
        shaped graph on one GPU: +17-20 %), for blocks large enough to be bound by the gathers; None otherwise."""
        from . import bigcsr

        if not bigcsr.ORDER_ROWS or not rowptr.is_cuda or colind.numel() < (1 << 24):
            return None
        return bigcsr.window_degree_order(rowptr)

    def gather(self, x, idx):
        from .pipeline import gather_rows_by_id

        return gather_rows_by_id(x.detach(), idx)

    def transpose(self, rowptr, colind, val, n_cols):
        from .plan import csr2csc, gather_rows

        plan = csr2csc(rowptr, colind, n_cols)
        return plan.colptr, plan.rowind, (gather_rows(plan.perm, val) if val is not None else None)


class HostBackend:
    """Local kernels on the HOST through libcogdl_host (CogDL's own CPU operator, spmm_cpu: operators/spmm.py:38) for
    CPU tensors over gloo: the launcher self-test of bench.py (`--selftest-cpu`) and CPU-resident shards.  Never chosen
    implicitly -- GPU shards without libcogdl_hip fail, they do not land here."""

    def spmm(self, rowptr, colind, val, x, out=None, order=None):
        from .operators.spmm import spmm_cpu

        y = spmm_cpu(rowptr.int(), colind.int(), val, x.detach().float())
        return y if out is None else out.add_(y)

    def transpose(self, rowptr, colind, val, n_cols):
        rows = torch.repeat_interleave(torch.arange(rowptr.numel() - 1), (rowptr[1:] - rowptr[:-1]).long())
        order = torch.sort(colind.long(), stable=True).indices  # stable: CSR order inside a column
        colptr = _csr_from_sorted_rows(colind.long()[order], n_cols).int()
        return colptr, rows[order].int(), (val[order] if val is not None else None)


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


def _raise_on_shard_flags(flags, n_global, what):
    """The flag word of cogdl_hip_shard_count (csrc/shard.hip): bit 0 = a column id outside [0, n_global), bit 1 = a
    row pointer that runs backwards, out of col[0, nnz), or does not end at nnz.  The kernel validates BEFORE it reads
    through the pointers (offending rows / columns are skipped), the host raises here."""
    from . import _lib

    if flags & 2:
        raise _lib.BackendError("%s: rowptr is not a non-decreasing CSR row pointer ending at nnz = colind.numel()" % what)
    if flags & 1:
        raise _lib.BackendError("%s: a column id lies outside [0, %d)" % (what, n_global))


def _check_csr_pointer(rp, nnz, what):
    """Host-side validation for the one-off preprocessing entry points whose kernels take no nnz (bfs_order): one
    reduction and one synchronisation per GRAPH, before any raw pointer is handed out."""
    from . import _lib

    if rp.numel() < 1 or int(rp[0]) != 0 or int(rp[-1]) != nnz or not bool((rp[1:] >= rp[:-1]).all()):
        raise _lib.BackendError("%s: rowptr must start at 0, be non-decreasing and end at colind.numel() = %d" % (what, nnz))


# ------------------------------------------------------------------------------------ the shard
class ShardedCSR:
    """One rank's shard of a row-partitioned CSR matrix plus its halo exchange plan."""

    def __init__(self, rowptr, colind_global, weight, bounds, group=None, backend=None):
        """rowptr [n_local+1] (any int dtype), colind_global [nnz] GLOBAL column ids (int64),
        weight [nnz] fp32 or None, bounds [world+1] (partition_bounds)."""
        from . import _lib

        self.group = group
        self.backend = backend or HipBackend()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = colind_global.device
        # A rank that raised here on its own would leave its peers blocked in the exchanges below: every rank validates
        # and splits locally -- ALL checks, the cheap shape checks included --, the ranks agree on the outcome with one
        # all-reduce, and then ALL of them raise or none.
        err = None
        try:
            bounds = bounds.to(device=dev, dtype=torch.long).contiguous()
            if bounds.numel() != self.world + 1:
                raise _lib.BackendError("ShardedCSR: %d partition bounds for %d ranks" % (bounds.numel(), self.world))
            blist = bounds.tolist()
            lo, hi = blist[self.rank], blist[self.rank + 1]
            self.n_local = hi - lo
            if rowptr.numel() != self.n_local + 1:
                raise _lib.BackendError("ShardedCSR: rowptr has %d entries, this rank owns %d rows" % (rowptr.numel(), self.n_local))
            if weight is not None and weight.numel() != colind_global.numel():
                raise _lib.BackendError("ShardedCSR: %d weights for %d edges" % (weight.numel(), colind_global.numel()))
            if colind_global.is_cuda and isinstance(self.backend, HipBackend):
                halo_ids, cut = self._split_hip(rowptr, colind_global, weight, bounds, lo, hi, blist[-1])
            else:  # CPU tensors (gloo tests, HostBackend): the same split with torch expressions
                _check_csr_pointer(rowptr.to(torch.long) - rowptr[0].to(torch.long), colind_global.numel(), "ShardedCSR")
                halo_ids, cut = self._split_torch(rowptr, colind_global, weight, bounds, lo, hi)
        except (_lib.BackendError, RuntimeError) as e:
            err = e
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
        if err is not None:
            raise err
        if int(bad.item()):
            raise _lib.BackendError("ShardedCSR: another rank rejected its shard (invalid row pointer / column ids); all ranks stop")
        self.n_halo = int(halo_ids.numel())
        self.nnz_local, self.nnz_remote = int(self.colind_loc.numel()), int(self.colind_rem.numel())
        # how many halo rows come from each owner, and which of MY rows each peer wants
        self.recv_counts = [int(cut[q + 1] - cut[q]) for q in range(self.world)]
        counts_t = torch.tensor(self.recv_counts, dtype=torch.long, device=dev)
        want_t, _ = exchange_rows(counts_t.view(-1, 1), [1] * self.world, [1] * self.world, group)
        self.send_counts = [int(v) for v in want_t.view(-1).tolist()]
        send_ids, _ = exchange_rows(halo_ids, self.recv_counts, self.send_counts, group)
        self.send_idx = (send_ids - lo).long()  # local row ids, grouped by requesting rank
        # Selection matrix of the backward accumulation: row r lists the positions j of `back` (= of send_idx) that
        # belong to local row r, in peer order (stable) -> gx += S . back is one deterministic csr_spmm_acc.
        if self.send_idx.is_cuda and isinstance(self.backend, HipBackend):
            from .graph_build import coo2csr_index

            try:  # the stable GPU COO -> CSR of the graph-construction path; raises on an id outside [0, n_local)
                sel_rowptr, order = coo2csr_index(self.send_idx, None, self.n_local)
            except _lib.BackendError as e:
                raise _lib.BackendError("ShardedCSR: a peer asked for a row this rank does not own (%s)" % e) from e
        else:
            if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_local):
                raise _lib.BackendError("ShardedCSR: a peer asked for a row this rank does not own")
            order = torch.sort(self.send_idx, stable=True).indices
            sel_rowptr = _csr_from_sorted_rows(self.send_idx[order], self.n_local)
        self.sel_rowptr = sel_rowptr.int()
        self.sel_colind = order.int()
        self._t_loc = self._t_rem = None
        self._comm = None  # explicit communication stream (GPU shards), created on first use

    def _split_hip(self, rowptr, colind_global, weight, bounds, lo, hi, n_global):
        """cogdl_hip_shard_count / cogdl_hip_shard_fill (csrc/shard.hip): two passes over the edges, int32 scratch, one
        host read of three sizes.  -> (halo_ids, cut)."""
        from . import _lib

        lib, dev = _lib.hip(), colind_global.device
        rowptr64, col64 = rowptr.to(torch.long).contiguous(), colind_global.to(torch.long).contiguous()
        w = None if weight is None else weight.to(torch.float32).contiguous()
        nnz = col64.numel()
        ws_bytes = lib.cogdl_hip_shard_workspace_bytes(self.n_local, n_global)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        counts = torch.empty(4, dtype=torch.long, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_shard_count(_lib.ptr(rowptr64), _lib.ptr(col64), self.n_local, nnz, lo, hi, n_global,
                                           _lib.ptr(counts), _lib.ptr(ws), ws_bytes, stream)
        _lib.check(rc, "shard_count")
        n_loc, n_rem, n_halo, flags = counts.tolist()  # the one synchronisation of building a shard
        _raise_on_shard_flags(flags, n_global, "ShardedCSR")

        def i32(n):
            return torch.empty(n, dtype=torch.int32, device=dev)

        self.rowptr_loc, self.rowptr_rem = i32(self.n_local + 1), i32(self.n_local + 1)
        self.colind_loc, self.colind_rem = i32(n_loc), i32(n_rem)
        self.w_loc = None if w is None else torch.empty(n_loc, dtype=torch.float32, device=dev)
        self.w_rem = None if w is None else torch.empty(n_rem, dtype=torch.float32, device=dev)
        halo_ids = torch.empty(n_halo, dtype=torch.long, device=dev)
        cut = torch.empty(self.world + 1, dtype=torch.long, device=dev)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_shard_fill(_lib.ptr(rowptr64), _lib.ptr(col64), _lib.ptr(w), self.n_local, nnz, lo, hi,
                                          n_global, _lib.ptr(bounds), self.world + 1, _lib.ptr(self.rowptr_loc),
                                          _lib.ptr(self.colind_loc), _lib.ptr(self.w_loc), _lib.ptr(self.rowptr_rem),
                                          _lib.ptr(self.colind_rem), _lib.ptr(self.w_rem), _lib.ptr(halo_ids), _lib.ptr(cut),
                                          _lib.ptr(ws), ws_bytes, stream)
        _lib.check(rc, "shard_fill")
        return halo_ids, cut.tolist()

    def _split_torch(self, rowptr, colind_global, weight, bounds, lo, hi):
        dev = colind_global.device
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
        return halo_ids, torch.searchsorted(halo_ids, bounds).tolist()

    # transposes for the backward pass, built on first use
    def transposed(self):
        if self._t_loc is None:
            self._t_loc = self.backend.transpose(self.rowptr_loc, self.colind_loc, self.w_loc, self.n_local)
            self._t_rem = self.backend.transpose(self.rowptr_rem, self.colind_rem, self.w_rem, self.n_halo)
        return self._t_loc, self._t_rem

    def schedule(self, which):
        """Row schedule of one of the shard's four blocks ("loc", "rem", "t_loc", "t_rem"), built on first use (plan time)."""
        memo = self.__dict__.setdefault("_schedules", {})
        if which not in memo:
            make = getattr(self.backend, "row_schedule", None)
            if make is None:
                memo[which] = None
            elif which == "loc":
                memo[which] = make(self.rowptr_loc, self.colind_loc)
            elif which == "rem":
                memo[which] = make(self.rowptr_rem, self.colind_rem)
            else:
                t = self.transposed()[0 if which == "t_loc" else 1]
                memo[which] = make(t[0], t[1])
        return memo[which]

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


def _spmm(be, rowptr, colind, w, x, out=None, order=None):
    """backend.spmm; the row schedule is passed only where there is one (backends of tests need not know the argument)."""
    if order is None:
        return be.spmm(rowptr, colind, w, x) if out is None else be.spmm(rowptr, colind, w, x, out=out)
    return be.spmm(rowptr, colind, w, x, out=out, order=order)


class _ShardedSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sh):
        be = sh.backend
        send = be.gather(x, sh.send_idx) if hasattr(be, "gather") else x.index_select(0, sh.send_idx)
        halo, done = _exchange_overlapped(sh, send, sh.send_counts, sh.recv_counts)
        y = _spmm(be, sh.rowptr_loc, sh.colind_loc, sh.w_loc, x, order=sh.schedule("loc"))  # overlaps with the all-to-all
        done()
        if sh.n_halo:
            y = _spmm(be, sh.rowptr_rem, sh.colind_rem, sh.w_rem, halo, out=y, order=sh.schedule("rem"))
        ctx.sh = sh
        return y

    @staticmethod
    def backward(ctx, g):
        sh = ctx.sh
        be = sh.backend
        g = g.contiguous()
        (cp_l, ri_l, w_l), (cp_r, ri_r, w_r) = sh.transposed()
        g_halo = _spmm(be, cp_r, ri_r, w_r, g, order=sh.schedule("t_rem")) if sh.n_halo else g.new_zeros((0, g.shape[1]))
        back, done = _exchange_overlapped(sh, g_halo, sh.recv_counts, sh.send_counts)
        gx = _spmm(be, cp_l, ri_l, w_l, g, order=sh.schedule("t_loc"))  # overlaps with the reverse all-to-all
        done()
        if back.shape[0]:
            gx = be.spmm(sh.sel_rowptr, sh.sel_colind, None, back, out=gx)  # gx += S . back: fixed order, one launch
        return gx, None


def sharded_spmm(sh, x_local):
    """Y_local = (A X)[my rows]; differentiable w.r.t. x_local."""
    return _ShardedSpMM.apply(x_local.contiguous(), sh)


# ---------------------------------------------------------------------------------- partitioner
def bfs_order(rowptr, colind, sources=None, max_levels=1 << 20):
    """Locality reordering for a contiguous 1-D partition: the vertices in breadth-first order (level by level, ties by
    id) from `sources` (default: vertex 0; every component that the search has not reached is then searched from its
    smallest vertex).  GPU graphs: cogdl_hip_bfs_step per level (csrc/shard.hip), the order itself as a stable transpose
    (cogdl_hip_csr2csc) of the vertex -> level map -- no torch sort.  -> perm (int64): perm[i] = the vertex that gets
    the new id i.  rowptr / colind: int64 CSR of a SYMMETRIC structure (what CogDL's preprocessing produces)."""
    from . import _lib
    from .plan import csr2csc

    dev = rowptr.device
    if not rowptr.is_cuda:
        raise _lib.BackendError("bfs_order: the graph must live on the GPU")
    n = rowptr.numel() - 1
    rp, ci = rowptr.to(torch.long).contiguous(), colind.to(torch.long).contiguous()
    _check_csr_pointer(rp, ci.numel(), "bfs_order")  # (cogdl_hip_bfs_step reads col[rowptr[u] .. rowptr[u+1]) unguarded)
    level = torch.full((n,), -1, dtype=torch.int32, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    src = torch.zeros(1, dtype=torch.long, device=dev) if sources is None else sources.to(dev).long()
    cur = 0
    lib = _lib.hip()
    stream = torch.cuda.current_stream(dev).cuda_stream
    lonely = (rp[1:] - rp[:-1]) == 0  # isolated vertices (the bulk of what an R-MAT generator leaves over): one last level
    if sources is None and n and bool(lonely[0]):
        first = torch.nonzero(~lonely)
        src = first[0] if first.numel() else src
    searches = 0
    while n and not bool(lonely.all()):
        searches += 1
        level[src] = cur
        while True:
            changed.zero_()
            with _lib.on_device(dev):
                _lib.check(lib.cogdl_hip_bfs_step(_lib.ptr(rp), _lib.ptr(ci), n, _lib.ptr(level), cur, _lib.ptr(changed), stream),
                           "bfs_step")
            cur += 1
            if not int(changed.item()) or cur >= max_levels:
                break
        rest = torch.nonzero((level < 0) & ~lonely)  # the next component, searched from its smallest vertex
        if rest.numel() == 0:
            break
        # (a graph of a million two-vertex components must not cost a million host round trips: after 32 single-source
        #  searches everything that is left starts at once)
        src = rest[0] if searches < 32 else rest.flatten()
    level[level < 0] = cur
    # order = vertices sorted by (level, id): the transpose of the n x n_levels matrix with one entry per vertex
    iota = torch.arange(n + 1, dtype=torch.int32, device=dev)
    plan = csr2csc(iota, level, cur + 1)
    return plan.rowind.long()


def edge_balanced_bounds(rowptr, world):
    """Contiguous row ranges with (nearly) equal EDGE counts -- equal row counts leave a power-law graph's first shard
    with several times the work of the last."""
    rp = rowptr.to(torch.long)
    n, nnz = rp.numel() - 1, int(rp[-1] - rp[0])
    targets = (torch.arange(1, world, dtype=torch.long, device=rp.device) * nnz) // world + rp[0]
    inner = torch.searchsorted(rp, targets).clamp(max=n)
    b = torch.cat([torch.zeros(1, dtype=torch.long, device=rp.device), inner, torch.tensor([n], dtype=torch.long, device=rp.device)])
    return torch.cummax(b, 0).values.cpu()


def permute_graph(rowptr, colind, weight, perm):
    """P A P^T: vertex perm[i] becomes vertex i (rows and columns).  GPU graphs: cogdl_hip_subgraph over ALL vertices in
    the new order (the induced-subgraph kernel relabels to positions in the list: exactly the permutation)."""
    from .operators.sample import subgraph_c

    rp, ci, _, edges = subgraph_c(rowptr.to(torch.long), colind.to(torch.long), perm)
    return rp, ci, (None if weight is None else weight[edges])


def halo_rows(rowptr, colind, bounds):
    """For every rank of the contiguous partition `bounds`: (edges to other ranks' columns, distinct such columns = rows of
    its halo table).  GPU graphs: cogdl_hip_shard_count per rank."""
    from . import _lib

    dev = rowptr.device
    lib = _lib.hip()
    rp, ci = rowptr.to(torch.long).contiguous(), colind.to(torch.long).contiguous()
    _check_csr_pointer(rp, ci.numel(), "halo_rows")  # (the per-rank slices below are cut with rp's own entries)
    blist = [int(v) for v in bounds.tolist()]
    n_global = blist[-1]
    out = []
    counts = torch.empty(4, dtype=torch.long, device=dev)
    for p in range(len(blist) - 1):
        lo, hi = blist[p], blist[p + 1]
        e0, e1 = int(rp[lo]), int(rp[hi])
        ws_bytes = lib.cogdl_hip_shard_workspace_bytes(hi - lo, n_global)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            rc = lib.cogdl_hip_shard_count(_lib.ptr(rp[lo:hi + 1]), _lib.ptr(ci[e0:e1]), hi - lo, e1 - e0, lo, hi, n_global,
                                           _lib.ptr(counts), _lib.ptr(ws), ws_bytes, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "shard_count")
        c = counts.tolist()
        _raise_on_shard_flags(c[3], n_global, "halo_rows")
        out.append((c[1], c[2]))
    return out


class Partition:
    """What partition() returns: the reordered graph, the permutation and what the reordering bought."""

    def __init__(self, rowptr, colind, weight, perm, bounds, halo_before, halo_after):
        self.rowptr, self.colind, self.weight, self.perm, self.bounds = rowptr, colind, weight, perm, bounds
        self.inverse = torch.empty_like(perm)
        self.inverse[perm] = torch.arange(perm.numel(), device=perm.device)
        self.halo_before, self.halo_after = halo_before, halo_after  # per rank: (remote edges, halo rows)

    def halo_fraction(self, which="after"):
        """Halo rows of the worst rank as a fraction of its own rows."""
        h = self.halo_after if which == "after" else self.halo_before
        rows = [int(self.bounds[p + 1] - self.bounds[p]) for p in range(len(h))]
        return max(hr / max(r, 1) for (_, hr), r in zip(h, rows))

    def shard(self, rank):
        """(rowptr, colind_global, weight) of rank's rows of the REORDERED graph: what ShardedCSR takes."""
        lo, hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        e0, e1 = int(self.rowptr[lo]), int(self.rowptr[hi])
        return (self.rowptr[lo:hi + 1] - self.rowptr[lo], self.colind[e0:e1],
                None if self.weight is None else self.weight[e0:e1])


def partition(rowptr, colind, world, weight=None, order="bfs", balance="edges"):
    """A 1-D vertex partition of a real (GPU-resident, symmetric) CSR graph for ShardedCSR: relabel for locality
    (`order`: "bfs" | "multilevel" (cogdl_amd/partitioner.py: label-propagation coarsening, greedy graph growing on the
    coarsest level, label-propagation refinement on the way back -- the parts ARE the ranks' ranges, breadth-first order
    inside each) | "degree" (hubs first) | "none"), cut the new id range into `world` contiguous ranges (`balance`:
    "edges" | "rows"; "multilevel" cuts at its part boundaries), and measure the halo of every rank before and after.  The analogue in the reference is the
    METIS partition of ClusteredDataset (cogdl/data/sampler.py:188-243), host-side and for sampling; here every step is
    a HIP kernel (bfs_step, csr2csc for the order, subgraph for the permutation, shard_count for the halos).
    x / y of the vertices follow with x[part.perm]; results come back with out[part.inverse]."""
    n = rowptr.numel() - 1
    dev = rowptr.device
    rp, ci = rowptr.to(torch.long), colind.to(torch.long)

    def cut(r):
        return edge_balanced_bounds(r, world) if balance == "edges" else partition_bounds(n, world)

    before = halo_rows(rp, ci, cut(rp))
    if order == "bfs":
        perm = bfs_order(rp, ci)
    elif order == "degree":
        from .plan import csr2csc

        deg = (rp[1:] - rp[:-1])
        key = (int(deg.max()) - deg).int()  # hubs first, ties by id: the stable transpose of vertex -> (max - degree)
        perm = csr2csc(torch.arange(n + 1, dtype=torch.int32, device=dev), key, int(deg.max()) + 1).rowind.long()
    elif order == "multilevel":
        from .partitioner import multilevel_partition
        from .plan import csr2csc

        labels = multilevel_partition(rp, ci, world)
        perm0 = bfs_order(rp, ci)  # inside a part: breadth-first order (neighbouring rows gather neighbouring columns)
        inner = csr2csc(torch.arange(n + 1, dtype=torch.int32, device=dev), labels[perm0].int(), world).rowind.long()
        perm = perm0[inner]        # by (part, breadth-first position)
        lp_bounds = torch.zeros(world + 1, dtype=torch.long, device=dev)
        lp_bounds[1:] = torch.cumsum(torch.bincount(labels, minlength=world), 0)
    elif order == "none":
        perm = torch.arange(n, device=dev)
    else:
        raise ValueError("partition: order must be 'bfs', 'multilevel', 'degree' or 'none'")
    if order == "none":
        rp2, ci2, w2 = rp, ci, weight
    else:
        rp2, ci2, w2 = permute_graph(rp, ci, weight, perm)
    bounds = lp_bounds.cpu() if order == "multilevel" else cut(rp2)
    after = halo_rows(rp2, ci2, bounds)
    return Partition(rp2, ci2, w2, perm, bounds, before, after)


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


def papers_graph_shard(rank, world, device, symmetrise=True, seed=0, num_nodes=None, num_pairs=None, bucket_edges=1 << 28):
    """This rank's rows of THE graph the one-GPU leg runs (bench.py `configs4_papers_1gpu`; synth.papers100m_like: seeded
    R-MAT pairs with ogbn-papers100M's node and pair counts, symmetrised as cogdl/datasets/ogb.py:50-55 feeds it to GCN,
    multi-edges kept, sym-normalised weights from the GLOBAL degrees) under a contiguous edge-balanced 1-D partition
    (edge_balanced_bounds).  Every rank generates the same pairs from the same seed and keeps the aggregation targets of
    its own row range -- nothing about the communication volume is an input: the halo table and the remote edge share come
    out of ShardedCSR as measurements.  -> (rowptr [n_local + 1] int64, colind [nnz_local] int32 GLOBAL ids, weight fp32,
    bounds [world + 1], nnz_global).  The same rows as synth.big_csr_from_pairs builds, cut at the bounds."""
    from . import synth

    num_nodes = synth.PAPERS_NODES if num_nodes is None else int(num_nodes)
    num_pairs = synth.PAPERS_PAIRS if num_pairs is None else int(num_pairs)
    dev = torch.device(device)
    src, dst = synth.rmat_pairs_i32(num_nodes, num_pairs, seed, dev)
    deg = torch.bincount(dst, minlength=num_nodes)
    if symmetrise:
        deg += torch.bincount(src, minlength=num_nodes)
    rowptr_g = torch.zeros(num_nodes + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=rowptr_g[1:])
    nnz_global = int(rowptr_g[-1])
    bounds = edge_balanced_bounds(rowptr_g, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    dinv = deg.to(torch.float32).pow_(-0.5)
    dinv[torch.isinf(dinv)] = 0
    del deg
    rowptr = (rowptr_g[lo:hi + 1] - rowptr_g[lo]).contiguous()
    del rowptr_g
    nnz = int(rowptr[-1])
    colind = torch.empty(nnz, dtype=torch.int32, device=dev)
    weight = torch.empty(nnz, dtype=torch.float32, device=dev)
    # my rows in buckets of about bucket_edges edges (cut by edge count: R-MAT rows are skewed towards low ids)
    n_b = max(1, (nnz + bucket_edges - 1) // bucket_edges)
    targets = torch.arange(1, n_b, device=dev, dtype=torch.int64) * (nnz // n_b)
    cuts = sorted(set([0, hi - lo] + [int(r) for r in (torch.searchsorted(rowptr, targets, right=True) - 1).tolist()]))
    for r0, r1 in zip(cuts, cuts[1:]):
        g0, g1 = lo + r0, lo + r1
        sel = (dst >= g0) & (dst < g1)
        key = dst[sel].long() * num_nodes + src[sel].long()
        if symmetrise:
            sel = (src >= g0) & (src < g1)
            key = torch.cat([key, src[sel].long() * num_nodes + dst[sel].long()])
        del sel
        key = torch.sort(key).values
        e0, e1 = int(rowptr[r0]), int(rowptr[r1])
        if key.numel() != e1 - e0:
            raise RuntimeError("papers_graph_shard: bucket [%d, %d) holds %d edges, the row pointer says %d" % (g0, g1, key.numel(), e1 - e0))
        row, col = key // num_nodes, key % num_nodes
        del key
        colind[e0:e1] = col
        weight[e0:e1] = dinv[row] * dinv[col]
        del row, col
    del src, dst, dinv
    if dev.type == "cuda":
        torch.cuda.empty_cache()
    return rowptr, colind, weight, bounds, nnz_global


def _value_at_world1():
    """The N = 1 point of the strong-scaling curve: fwd+bwd GEdges/s of the symmetrised papers100M-shaped graph on ONE GPU
    (bench.py `configs4_papers_1gpu`, tools/papers_bench.py), from the newest committed profile -- a number measured by an
    earlier one-GPU run of the same code path, repeated here so that a curve has an origin; the driver's own N = 1 run
    carries the live value in its `configs4_papers_1gpu.symmetrised.forward_backward.GEdges_s`."""
    import glob
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in sorted(glob.glob(os.path.join(root, "profiles", "r*_papers_1gpu.json")), reverse=True):
        try:
            runs = json.load(open(path)).get("runs") or []
            fb = runs[-1]["symmetrised"]["forward_backward"]
            return {"GEdges_s": fb["GEdges_s"], "ms_per_step": fb["ms"], "nnz": runs[-1]["symmetrised"]["nnz"],
                    "source": "profiles/%s (committed one-GPU measurement; NOT measured in this run)" % os.path.basename(path)}
        except (OSError, KeyError, IndexError, ValueError):
            continue
    return None


def _rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return None


def sharded_leg(rank, world, dev, shard_nodes, degree, feat, remote_frac, halo_frac, steps, warmup, backend=None, seed=0,
                dump_dir=None, shard=None):
    """One measured leg of the N > 1 bench, run by EVERY rank of an initialised process group: take this rank's shard --
    `shard` = (rowptr, colind_global, weight, bounds, nnz_global) of a real graph cut by its own partition
    (papers_graph_shard), or None: generate a papers100M-shaped shard from the knobs (remote_frac, halo_frac) --, build the
    exchange plan, time `steps` forward + backward passes (barrier + synchronise on both sides, max over ranks), then the
    local block alone.  Returns the same dict on every rank."""
    cuda = dev.type == "cuda"

    def sync():
        if cuda:
            torch.cuda.synchronize(dev)

    if shard is None:
        rowptr, cols, w = _papers_like_shard(rank, world, shard_nodes, degree, remote_frac, seed, dev, halo_frac)
        bounds = torch.arange(world + 1, dtype=torch.long) * shard_nodes
    else:
        rowptr, cols, w, bounds = shard[:4]
    n_local = int(rowptr.numel()) - 1
    lo = int(bounds[rank])
    sync()
    t_plan = time.perf_counter()
    sh = ShardedCSR(rowptr, cols, w, bounds, backend=backend)
    sync()
    t_plan = time.perf_counter() - t_plan
    x = torch.randn(n_local, feat, device=dev, requires_grad=True)
    gout = torch.randn(n_local, feat, device=dev)
    if dump_dir:  # launcher self-test: this rank's shard, operands and one forward + backward, for the test's oracle
        import numpy as np

        y = sharded_spmm(sh, x)
        y.backward(gout)
        np.savez(os.path.join(dump_dir, "b%d.npz" % rank), rowptr=rowptr.cpu().numpy(), cols=cols.cpu().numpy(),
                 w=w.cpu().numpy(), x=x.detach().cpu().numpy(), gout=gout.cpu().numpy(), y=y.detach().cpu().numpy(),
                 gx=x.grad.cpu().numpy(), n_halo=sh.n_halo, nnz_remote=sh.nnz_remote, lo=lo, n_local=n_local)
        x.grad = None
        del y
    del cols, rowptr, w, shard
    if cuda:
        torch.cuda.empty_cache()

    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # (gloo: the small collectives on host tensors)

    def allsum(v, op=dist.ReduceOp.SUM, dtype=torch.float64):
        t = torch.tensor([v], device=cdev, dtype=dtype)
        dist.all_reduce(t, op=op)
        return t.item()

    def allgather(v, dtype=torch.float64):
        out = [torch.zeros(1, dtype=dtype, device=cdev) for _ in range(world)]
        dist.all_gather(out, torch.tensor([v], dtype=dtype, device=cdev))
        return [t.item() for t in out]

    nnz_global = int(allsum(sh.nnz_local + sh.nnz_remote, dtype=torch.long))

    def step():
        y = sharded_spmm(sh, x)
        x.grad = None
        y.backward(gout)

    for _ in range(warmup):
        step()
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dist.barrier()
    dt = allsum(time.perf_counter() - t0, op=dist.ReduceOp.MAX)
    # the local block alone (no exchange): what one GPU does on its own columns -- the dominant kernel of a step
    with torch.no_grad():
        for _ in range(2):
            _spmm(sh.backend, sh.rowptr_loc, sh.colind_loc, sh.w_loc, x, order=sh.schedule("loc"))
        sync()
        reps = 10 if cuda else 2
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t1 = time.perf_counter()
        for _ in range(reps):
            _spmm(sh.backend, sh.rowptr_loc, sh.colind_loc, sh.w_loc, x, order=sh.schedule("loc"))
        if cuda:
            e1.record()
        sync()
        loc_ms = (e0.elapsed_time(e1) if cuda else (time.perf_counter() - t1) * 1e3) / reps
    loc_all = [float(v) for v in allgather(loc_ms)]
    halo_by_rank = [int(v) for v in allgather(sh.n_halo, torch.long)]
    rows_by_rank = [int(v) for v in allgather(n_local, torch.long)]
    edges_by_rank = [int(v) for v in allgather(sh.nnz_local + sh.nnz_remote, torch.long)]
    remote_by_rank = [int(v) for v in allgather(sh.nnz_remote, torch.long)]
    halo_rows = sum(halo_by_rank)
    remote_edges = sum(remote_by_rank)
    b_alg = sh.nnz_local * (4 + 4 + feat * 4) + n_local * (4 + feat * 4)
    nnz_gpu = nnz_global / world
    out = {
        "value": 2 * nnz_global * steps / dt / 1e9, "unit": "GEdges/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
        "warmup": warmup, "nodes_per_gpu": n_local, "nnz_global": nnz_global, "feat": feat,
        "remote_frac": remote_frac, "halo_frac": halo_frac, "remote_edge_share": remote_edges / max(nnz_global, 1),
        "halo_rows_all_ranks": int(halo_rows), "halo_rows_rank0": sh.n_halo,
        "halo_rows_by_rank": halo_by_rank, "rows_by_rank": rows_by_rank, "edges_by_rank": edges_by_rank,
        "remote_edges_by_rank": remote_by_rank,
        "halo_GB_per_step_all_ranks": halo_rows * feat * 4 * 2 / 1e9,
        "exchange_plan_build_s_rank0": t_plan,
        "local_block_ms_by_rank": [round(v, 3) for v in loc_all],
        "local_block_ms_min": min(loc_all), "local_block_ms_max": max(loc_all),
        "local_block_spmm_ms_rank0": loc_all[0],
        "local_block_GEdges_s_rank0": sh.nnz_local / (loc_all[0] * 1e-3) / 1e9,
        "local_block_algorithmic_bytes": b_alg,
    }
    if shard_nodes:  # the knob-generated shards: a PRIOR for their curve (predict_scaling); a real graph's line carries none
        # (the per-rank halo of the bench's generator: halo_frac x the shard's rows wherever ranks have peers; measured when they do)
        halo_pred = sh.n_halo if world > 1 else (halo_frac * shard_nodes if halo_frac > 0 else min(remote_frac * nnz_gpu, 7.0 * shard_nodes))
        rf_pred = remote_frac if remote_frac >= 0 else 0.1
        out["predicted"] = predict_scaling(shard_nodes, nnz_gpu, feat, rf_pred if world == 1 else remote_edges / max(nnz_global, 1),
                                           halo_pred, max(loc_all) / max(sh.nnz_local / 1e9, 1e-12))
    return out


XGMI_LINK_GBS = 153.0   # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point)
XGMI_LINK_EFF = 0.8     # what a large RCCL send/recv is ASSUMED to reach of it -- a stated prior, not a measurement


def predict_scaling(shard_nodes, nnz_per_gpu, feat, remote_frac, halo_rows_per_rank, local_ms_per_gedge, worlds=(2, 4, 8),
                    link_gbs=XGMI_LINK_GBS, link_eff=XGMI_LINK_EFF, elem=4):
    """A falsifiable PRIOR for the weak-scaling curve of the sharded SpMM (forward + backward), made from what ONE GPU can
    measure -- so that the first hardware 1 -> 2 -> 4 -> 8 run has something to be compared with.

    Model of one step on every rank (all ranks alike), the structure of `_ShardedSpMM`:
        forward :  gather  ->  [ all-to-all of the halo rows  ||  local-block SpMM ]  ->  halo-block SpMM
        backward:  halo-block transpose SpMM  ->  [ reverse all-to-all  ||  local-block transpose SpMM ]  ->  accumulate
      t_local  = (1 - remote_frac) * nnz_per_gpu edges at the measured rate of the local block (ms per 10^9 edges)
      t_remote = remote_frac * nnz_per_gpu edges at the same rate (the halo table is smaller than X: an upper bound)
      t_a2a    = halo_rows * feat * elem bytes per direction, spread evenly over the N - 1 peers = N - 1 xGMI links used
                 at once (point to point: no switch), each at link_gbs * link_eff
      t_fix    = gather of the requested rows + accumulation of the returned ones: 3 passes over halo_rows * feat * elem at
                 ~2 TB/s (row gathers at mini-batch granularity)
      step     = 2 * max(t_local, t_a2a) + 2 * t_remote + t_fix;     efficiency(N) = step(1) / step(N),
      step(1)  = 2 * nnz_per_gpu edges at the measured rate (world size 1: every edge is local, no exchange).
    What it deliberately leaves out -- and the hardware run will show -- is RCCL's launch / protocol latency per
    all-to-all (tens of microseconds against ~10-40 ms here), contention between the exchange and the local block for
    HBM, and load imbalance between ranks (max over ranks, not the mean)."""
    out = {"model": "step = 2*max(t_local, t_a2a) + 2*t_remote + t_fix; a2a over the N-1 point-to-point xGMI links at "
                    "%.0f GB/s x %.2f each; local/remote blocks at the measured single-GPU rate" % (link_gbs, link_eff),
           "inputs": {"shard_nodes": shard_nodes, "nnz_per_gpu": nnz_per_gpu, "feat": feat, "remote_frac": remote_frac,
                      "halo_rows_per_rank": halo_rows_per_rank, "local_ms_per_gedge": local_ms_per_gedge}}
    gedges = nnz_per_gpu / 1e9
    step1 = 2 * gedges * local_ms_per_gedge
    halo_bytes = halo_rows_per_rank * feat * elem
    for n in worlds:
        t_local = (1 - remote_frac) * gedges * local_ms_per_gedge
        t_remote = remote_frac * gedges * local_ms_per_gedge
        t_a2a = halo_bytes / max(n - 1, 1) / (link_gbs * 1e9 * link_eff) * 1e3
        t_fix = 3 * halo_bytes / 2e12 * 1e3
        step = 2 * max(t_local, t_a2a) + 2 * t_remote + t_fix
        out[str(n)] = {"halo_GB_per_rank_per_direction": round(halo_bytes / 1e9, 4), "a2a_ms": round(t_a2a, 3),
                       "local_block_ms": round(t_local, 3), "remote_block_ms": round(t_remote, 3), "fixed_ms": round(t_fix, 3),
                       "step_ms": round(step, 3), "exchange_hidden": bool(t_a2a <= t_local),
                       "efficiency": round(step1 / step, 4),
                       "GEdges_s_all_gpus": round(2 * n * gedges / (step * 1e-3), 2)}
    out["step_ms_world1"] = round(step1, 3)
    return out


def _child_leg(argv, port_offset, timeout_s):
    """Run a follow-up leg (`python <argv>`) as a CHILD interpreter of this rank, with the parent's RANK / LOCAL_RANK /
    WORLD_SIZE and a MASTER_PORT of its own: the children of all ranks form their own process group.  A leg that hangs
    or dies is killed at `timeout_s` and reported as an error -- it cannot take the main measurement down with it."""
    import json
    import subprocess
    import sys

    # (TORCHELASTIC_USE_AGENT_STORE would make the child's env:// rendezvous look for the launcher's store at the new
    # port instead of letting its rank 0 create one)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29533")) + port_offset)
    env["COGDL_AMD_BENCH_CHILD"] = "1"
    try:
        proc = subprocess.run([sys.executable] + argv, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout_s}
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        if proc.returncode == 0 and int(env.get("RANK", "0")) != 0:
            return {}  # only rank 0 of a leg prints its line: a clean exit without one is what the other ranks look like
        return {"error": "rc %d: %s" % (proc.returncode, (proc.stderr or proc.stdout)[-400:])}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": "unparsable output: %r" % (e,)}


def _any_rank(flag, port_offset, timeout_s=120):
    """True on EVERY rank iff `flag` is true on at least one of them -- the ranks' vote on whether a follow-up leg has to be
    repeated in another form, through a TCPStore of its own (the main process group is gone by then; a rank deciding on its
    own would wait alone in the next leg's rendezvous until the timeout).  None when the vote itself fails."""
    from datetime import timedelta

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world == 1:
        return bool(flag)
    port = int(os.environ.get("MASTER_PORT", "29533")) + port_offset
    try:
        store = dist.TCPStore("127.0.0.1", port, world, rank == 0, timeout=timedelta(seconds=timeout_s))
        store.add("yes", 1 if flag else 0)
        store.add("in", 1)
        deadline = time.time() + timeout_s
        while store.add("in", 0) < world:
            if time.time() > deadline:
                return None
            time.sleep(0.01)
        verdict = store.add("yes", 0) > 0
        store.add("out", 1)
        while rank == 0 and store.add("out", 0) < world and time.time() < deadline:  # the store lives in rank 0
            time.sleep(0.01)
        return verdict
    except Exception:
        return None


def bench_sharded_spmm(args):
    """Weak-scaling bench of the vertex-sharded csr_spmm forward + backward (bench.py --gpus N, N > 1; also the
    `weak_scaling_base` leg of the N = 1 line: world size 1).  Every rank runs this; rank 0 returns the result."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", rank))
    cpu = bool(getattr(args, "selftest_cpu", False))
    leg = getattr(args, "leg", "main")
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: refusing to report a %d-GPU line from %d "
                         "rank(s)" % (args.gpus, world, args.gpus, world))
    share = bool(getattr(args, "share_gpu", False))
    if cpu:  # launcher self-test (tests/test_dist_cpu.py): gloo ranks on the host, libcogdl_host kernels, tiny shards
        dev, backend_name, backend = torch.device("cpu"), "gloo", HostBackend()
    elif share:  # orchestration smoke test on a ONE-GPU box: every rank on cuda:0, rows exchanged through gloo (RCCL refuses
        dev, backend_name, backend = torch.device("cuda", 0), "gloo", None  # two ranks per device); HIP kernels throughout
        torch.cuda.set_device(dev)
    else:
        n_dev = torch.cuda.device_count()
        if local >= n_dev:
            raise SystemExit("bench.py: rank %d needs device %d, this host has %d GPU(s)" % (rank, local, n_dev))
        dev, backend_name, backend = torch.device("cuda", local), "nccl", None
        torch.cuda.set_device(dev)
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if cpu or share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_ranks_seen = dist.get_world_size()
    # Which graph (round-5 verdict, star-g3).  The MAIN leg at N > 1 shards THE graph the N = 1 line's `configs4_papers_1gpu`
    # leg runs -- the papers100M-shaped symmetrised graph, every rank keeping the rows of its edge-balanced contiguous range
    # (papers_graph_shard): fixed total work = STRONG scaling, halo rows and remote edge share are measured outputs.  The
    # knob-generated shards (one papers100M/8-sized shard per GPU whose remote fraction and halo ratio are INPUTS: what a
    # locality-preserving partition is assumed to leave) are the follow-up leg `assumed_partition`, the world-size-1 base
    # (`weak_scaling_base` of the N = 1 line) and the worst-case leg.
    graph_leg = leg == "main" and world > 1
    f = args.feat
    shard_nodes = degree = remote_frac = halo_frac = None
    shard, scale = None, 1
    if graph_leg:
        from . import synth

        scale = int(getattr(args, "papers_scale", 0) or (2048 if cpu else 64 if share else 1))
        nodes, pairs = synth.PAPERS_NODES // scale, synth.PAPERS_PAIRS // scale
        t_gen = time.perf_counter()
        shard = papers_graph_shard(rank, world, dev, True, 0, nodes, pairs, bucket_edges=max(1 << 16, (1 << 28) // scale))
        t_gen = time.perf_counter() - t_gen
        bounds_list = [int(v) for v in shard[3].tolist()]
        m = sharded_leg(rank, world, dev, 0, 0.0, f, -1.0, -1.0, args.steps, args.warmup, backend, shard=shard,
                        dump_dir=os.environ.get("COGDL_AMD_SELFTEST_DUMP") if cpu else None)
        shard = None
    else:
        shard_nodes = args.shard_nodes or 111_059_956 // 8
        degree = args.shard_degree or 28.8                         # 3.2e9 symmetrised edges / 111e6 nodes
        remote_frac = args.remote_frac if args.remote_frac >= 0 else 0.1
        halo_frac = getattr(args, "halo_frac", 0.25)
        m = sharded_leg(rank, world, dev, shard_nodes, degree, f, remote_frac, halo_frac, args.steps, args.warmup, backend)
    rccl = None if cpu else _rccl_version()
    dist.barrier()
    t_start = float(getattr(args, "t0", _T_IMPORT))
    elapsed = torch.tensor([time.time() - t_start], dtype=torch.float64, device=dev if backend_name == "nccl" else "cpu")
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # the same figure on every rank: the legs below are skipped by ALL or none
    elapsed = float(elapsed)
    if own_group:
        dist.destroy_process_group()
    if not cpu:
        torch.cuda.empty_cache()
    result = None
    if rank == 0 and graph_leg:
        result = {
            "metric": "SpMM GEdges/s (vertex-sharded csr_spmm fwd+bwd on the papers100M-shaped graph, symmetrised) @%d GPUs" % world,
            "value": m["value"], "unit": "GEdges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: the papers100M-shaped graph of the N = 1 line's configs4_papers_1gpu leg "
                                   "(seeded R-MAT pairs, %d nodes, symmetrised as cogdl/datasets/ogb.py:50-55 feeds it to GCN, "
                                   "multi-edges kept where the reference coalesces, sym-normalised weights), rows cut into %d "
                                   "contiguous edge-balanced ranges, 1-D vertex-sharded csr_spmm fwd+bwd, F = %d fp32; halo "
                                   "rows and remote edge share are MEASURED on this partition, not inputs%s"
                                   % (synth.PAPERS_NODES // scale, world, f, "" if scale == 1 else " -- at 1/%d scale" % scale),
                       "nodes": synth.PAPERS_NODES // scale, "nnz_global": m["nnz_global"], "feat": f, "scale": scale,
                       "bounds": bounds_list, "rows_by_rank": m["rows_by_rank"], "edges_by_rank": m["edges_by_rank"],
                       "halo_rows_by_rank": m["halo_rows_by_rank"], "halo_rows_rank0": m["halo_rows_rank0"],
                       "remote_edges_by_rank": m["remote_edges_by_rank"], "remote_edge_share": m["remote_edge_share"],
                       "partition": "contiguous row ranges of (nearly) equal edge count (dist.edge_balanced_bounds), ids as generated",
                       "parallelism": "vertex-shard x%d, RCCL all-to-all halo exchange overlapped with local SpMM" % world},
            "graph_build_s_rank0": t_gen,
        }
        w1 = _value_at_world1()
        if w1 is not None and scale == 1:
            result["value_at_world1"] = w1
    elif rank == 0:
        result = {
            "metric": "SpMM GEdges/s (vertex-sharded csr_spmm fwd+bwd, papers100M-shaped shards) @%d GPUs" % world,
            "value": m["value"], "unit": "GEdges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "papers100M-like 1-D vertex-sharded csr_spmm fwd+bwd (configs[4]) on GENERATED shards, one "
                                   "papers100M/8-sized shard per GPU (locality-preserving partition, ASSUMED): %.1f%% of every "
                                   "row's sources in other shards, %s" % (100 * remote_frac, (
                                       "drawn from boundary regions sized for a halo of %.2f x the shard's rows "
                                       "(the worst_case_partition object is the other end: a random partition)" % halo_frac)
                                       if halo_frac > 0 else
                                       "uniform over the owners' rows (random partition of a structureless graph: no halo "
                                       "reuse, the worst case)"),
                       "nodes_per_gpu": shard_nodes, "nnz_global": m["nnz_global"], "feat": f, "remote_frac": remote_frac,
                       "halo_frac": halo_frac, "halo_rows_rank0": m["halo_rows_rank0"],
                       "parallelism": "vertex-shard x%d, RCCL all-to-all halo exchange overlapped with local SpMM" % world},
            # a PRIOR for the curve of THESE shards, from this run's own single-GPU rates and a stated xGMI rate
            # (predict_scaling): a model fed with the knobs above -- not evidence
            "predicted": m["predicted"],
        }
        if world > 1 and str(world) in m["predicted"]:
            pr = m["predicted"][str(world)]
            result["predicted_vs_measured"] = {"predicted_step_ms": pr["step_ms"], "measured_step_ms": m["ms_per_step"],
                                               "measured_over_predicted": m["ms_per_step"] / max(pr["step_ms"], 1e-9)}
    if rank == 0:
        result.update({
            "halo_GB_per_step_all_ranks": m["halo_GB_per_step_all_ranks"],
            "n_ranks_seen": n_ranks_seen, "rccl_version": rccl,
            "exchange": "all_to_all_single on an explicit comm stream (event-ordered), gather = cogdl_hip_gather_feature_rows, "
                        "backward accumulation = one csr_spmm_acc over the selection matrix",
            "exchange_plan_build_s_rank0": m["exchange_plan_build_s_rank0"],
            "local_block_ms_by_rank": m["local_block_ms_by_rank"],
            "local_block_ms_min": m["local_block_ms_min"], "local_block_ms_max": m["local_block_ms_max"],
            "local_block_spmm_ms_rank0": m["local_block_spmm_ms_rank0"],
            "local_block_GEdges_s_rank0": m["local_block_GEdges_s_rank0"],
        })
        if cpu:
            result["selftest"] = "gloo ranks on the host, libcogdl_host kernels: exercises the launcher and the data flow only"
        if share:
            result["selftest"] = ("all ranks on cuda:0, halo rows staged through gloo: exercises the orchestration (launcher, "
                                  "HIP shard construction, child legs) on a one-GPU box; its rate is not a multi-GPU number")
        # the dominant kernel of a step: csr_spmm over the rank's local block (SURVEY.md section 8d's formula), timed
        # above with HIP events on its own; X (7.1 GB per shard) is far beyond the caches, so this IS HBM traffic
        b_alg = m["local_block_algorithmic_bytes"]
        ach = b_alg / (m["local_block_ms_max"] * 1e-3) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": "rowreduce_main_kernel<SpmmOp<float,...>> on the local block A_pp "
                                                        "(the SLOWEST rank's launch time)",
                              "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                              "algorithmic_bytes_per_launch": b_alg}
    if leg != "main" or world == 1 or getattr(args, "no_extra_legs", False):
        return result
    # ---- follow-up legs, each in child interpreters with their own process group and a hard timeout -----------------
    # The whole command has to "finish within minutes" (the bench contract): every leg has a hard timeout, and a leg is
    # not started at all once the budget is gone -- decided on figures every rank shares (the all-reduced time above,
    # then a vote), never by a rank on its own.
    budget_s = float(getattr(args, "legs_budget_s", 420.0))
    if elapsed > budget_s:
        if rank == 0:
            result["legs_skipped"] = "main leg took %.0f s of a %.0f s budget" % (elapsed, budget_s)
        return result
    script = os.path.abspath(getattr(args, "bench_script", "bench.py"))
    common = ["--gpus", str(world), "--feat", str(f)] + (["--selftest-cpu"] if cpu else []) + (["--share-gpu"] if share else [])
    if shard_nodes is None:  # (the graph leg: the generated-shard legs below take their sizes from the arguments)
        shard_nodes = args.shard_nodes or 111_059_956 // 8
        degree = args.shard_degree or 28.8
    assumed = _child_leg([script, "--sharded", "--leg", "assumed"] + common
                         + ["--shard-nodes", str(shard_nodes), "--shard-degree", str(degree),
                            "--remote-frac", repr(args.remote_frac if args.remote_frac >= 0 else 0.1),
                            "--halo-frac", repr(getattr(args, "halo_frac", 0.25)),
                            "--steps", str(max(2, args.steps // 2)), "--warmup", "1"], 6, 300) if graph_leg else None
    worst_nodes = max(64, shard_nodes // max(1, int(getattr(args, "worst_case_scale", 4))))
    worst = _child_leg([script, "--sharded", "--leg", "worst"] + common
                       + ["--shard-nodes", str(worst_nodes), "--shard-degree", str(degree),
                          "--remote-frac", repr((world - 1) / world), "--halo-frac", "0",
                          "--steps", str(max(2, args.steps // 2)), "--warmup", "1"], 1, 240)
    sage = None
    run_sage = not getattr(args, "no_sage", False) and not share
    if run_sage and _any_rank(time.time() - t_start > budget_s, 5, 60) is not False:
        run_sage = False
        if rank == 0:
            result["legs_skipped"] = "configs3_sage_replicas: the %.0f s budget was spent (or the ranks' vote failed)" % budget_s
    if run_sage:
        tool = os.path.join(os.path.dirname(script), "tools", "sage_bench.py")
        captured, eager = [tool, "--captured", "--batch", "1024", "--steps", "50"], [tool, "--batch", "1024", "--steps", "30"]
        if cpu:  # launcher self-test: stand-ins that exercise the legs' control flow -- the first form fails on rank 1 ONLY
            captured = ["-c", "import os, sys; r = int(os.environ['RANK']); print('{\"selftest_leg\": \"captured\"}' if r == 0 "
                              "else ''); sys.exit(1 if r == 1 else 0)"]
            eager = ["-c", "import os; print('{\"selftest_leg\": \"eager\"}' if os.environ['RANK'] == '0' else '')"]
        sage = _child_leg(captured, 2, 180)
        # the RCCL all-reduce as a node of the captured graph failed somewhere: the eager step with torch DDP -- on ALL
        # ranks or on none (the ranks vote; only rank 0 of a leg prints a line, so a rank cannot tell from its own child)
        again = _any_rank("error" in sage, 4, 60)
        if again:
            first = sage
            sage = _child_leg(eager, 3, 180)
            sage["captured_attempt"] = first
        elif again is None:
            sage.setdefault("note", "the ranks' vote on repeating this leg failed; not repeated")
    if rank == 0:
        if "error" not in worst:
            worst = {k: worst[k] for k in ("value", "unit", "ms_per_step", "steps", "n_gpus", "n_ranks_seen", "config",
                                           "halo_GB_per_step_all_ranks", "local_block_ms_min", "local_block_ms_max")
                     if k in worst}
            worst["what"] = ("the same sharded fwd+bwd on a RANDOM partition of a structureless graph: (N-1)/N of every row's "
                             "sources remote and uniform over the owners (no halo reuse) -- the worst case of a 1-D "
                             "partition, on shards 1/%d the size so that the halo tables fit"
                             % max(1, int(getattr(args, "worst_case_scale", 4))))
        result["worst_case_partition"] = worst
        if assumed is not None:
            if "error" not in assumed:
                assumed = {k: assumed[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "n_gpus", "n_ranks_seen",
                                                   "scaling", "config", "halo_GB_per_step_all_ranks", "local_block_ms_min",
                                                   "local_block_ms_max", "predicted", "predicted_vs_measured") if k in assumed}
                assumed["what"] = ("the same sharded fwd+bwd on GENERATED shards (one papers100M/8-sized shard per GPU: weak "
                                   "scaling) whose remote fraction and halo ratio are inputs -- what a locality-preserving "
                                   "(METIS-like) partition is ASSUMED to leave; `predicted` is a model fed with those knobs")
            result["assumed_partition"] = assumed
        if sage is not None:
            result["configs3_sage_replicas"] = sage
    return result
