"""The mini-batch input pipeline of sampled training on the GPU.

The reference samples on the CPU inside DataLoader workers (cogdl/data/sampler.py:62-116 -> Graph.sample_adj ->
sample.cpp, single-threaded per worker), gathers `x[n_id]` on the host into a fresh tensor and copies every batch to
the GPU (cogdl/models/nn/graphsage.py:86-99).  Here, for a graph whose structure lives in HBM:

  gather_rows_by_id(src, ids)      x[n_id] as ONE kernel that reads the selected rows wherever they live -- HBM, or
                                   PINNED host memory read straight over the host link (zero copy: no host-side
                                   index_select, no staging buffer, no separate H2D copy);
  sample_blocks_padded /           the same sampling into buffers of FIXED capacity (no size depends on what was
  CapturedMiniBatchStep            sampled, nothing synchronises) and, on top of it, the whole training step -- sampling,
                                   gather, forward, backward, optimizer -- captured once and replayed as one hipGraph;
  BatchPipeline                    sampling (cogdl_hip_sample_adj per hop) + the feature gather of batch i+1 run on a
                                   side stream while batch i trains on the caller's stream;
  layerwise_inference              Graphsage.inference (graphsage.py:106-119): layer by layer over ALL nodes with full
                                   neighbourhoods (sample_adj(-1) on the GPU), features of the current layer gathered
                                   by the same kernel.
"""
import torch

from . import _lib, graphs
from . import plan as _plan
from .operators.sample import sample_adj_c, sample_adj_padded


GATHER_BAD_ID = 1 << 32  # bit set in a caller's 64-bit flag word by gather_rows_by_id(flag_word=...)


def gather_rows_by_id(src, ids, out=None, flag_word=None):
    """out[i] = src[ids[i]] along dim 0.  `src`: a CUDA tensor, or a PINNED CPU tensor (read by the GPU in place);
    `ids`: CUDA int64 / int32.  Returns a CUDA tensor on ids' device (stream-ordered on its current stream).
    An id outside the source is reported, not read: by default in a fresh flag (zeroed by one fill launch; check_gather
    reads it), or -- `flag_word`, an int64 device scalar the caller resets itself, e.g. the sampler's flags of the same
    step -- as bit GATHER_BAD_ID of that word (no fill launch: a captured step pays per kernel node)."""
    if not ids.is_cuda:
        raise _lib.BackendError("gather_rows_by_id: ids must live on the GPU (got %s)" % ids.device)
    dev = ids.device
    if src.is_cuda:
        if src.device != dev:
            raise _lib.BackendError("gather_rows_by_id: src on %s, ids on %s" % (src.device, dev))
    elif not src.is_pinned():
        raise _lib.BackendError("gather_rows_by_id: a host-resident source must be pinned (src.pin_memory()) so that the "
                                "GPU can read it in place; there is no staged-copy fallback")
    if not src.is_contiguous():
        raise _lib.BackendError("gather_rows_by_id: src must be contiguous")
    if ids.dtype not in (torch.int64, torch.int32) or ids.dim() != 1:
        raise _lib.BackendError("gather_rows_by_id: ids must be a 1-D int64/int32 tensor")
    ids = ids.contiguous()
    n, n_src = ids.numel(), src.shape[0]
    row_elems = src.numel() // max(n_src, 1) if n_src else 0
    row_bytes = row_elems * src.element_size()
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
    elif out.shape != (n,) + tuple(src.shape[1:]) or out.dtype != src.dtype or not out.is_contiguous() or out.device != dev:
        raise _lib.BackendError("gather_rows_by_id: `out` must be a contiguous %s tensor of shape %s on %s"
                                % (src.dtype, (n,) + tuple(src.shape[1:]), dev))
    if n == 0 or row_bytes == 0:
        return out
    if row_bytes % 4:
        raise _lib.BackendError("gather_rows_by_id: rows of %d bytes (must be a multiple of 4)" % row_bytes)
    if flag_word is None:
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        flag_ptr = _lib.ptr(bad)
    else:
        if (not torch.is_tensor(flag_word) or flag_word.dtype != torch.long or flag_word.device != dev
                or flag_word.numel() != 1):
            raise _lib.BackendError("gather_rows_by_id: flag_word must be an int64 scalar tensor on %s" % dev)
        bad, flag_ptr = None, flag_word.data_ptr() + 4  # the high half of the little-endian word: bit 32 of its value
    fn = _lib.hip().cogdl_hip_gather_feature_rows if ids.dtype == torch.int64 else _lib.hip().cogdl_hip_gather_feature_rows_i32
    with _lib.on_device(dev):
        rc = fn(_lib.ptr(ids), src.data_ptr(), _lib.ptr(out), n, row_bytes, n_src, flag_ptr, _lib.stream_of(ids))
    _lib.check(rc, "gather_feature_rows")
    if bad is not None:
        out._cogdl_bad_flag = bad  # checked lazily (BatchPipeline / tests): reading it here would stall the stream
    return out


def check_gather(out):
    """Raise if the gather that produced `out` met an id outside the source (synchronises)."""
    bad = getattr(out, "_cogdl_bad_flag", None)
    if bad is not None and int(bad.item()):
        raise _lib.BackendError("gather_rows_by_id: an id lies outside the source's rows")


def sample_blocks(indptr, indices, seeds, fanouts):
    """NeighborSampler.sample (cogdl/data/sampler.py:93-116) on the GPU: one sample_adj per hop, outermost hop last;
    returns (n_id, [((row_ptr, col), n_dst), ...] innermost block first)."""
    adjs = []
    batch = seeds
    for k in fanouts:
        row_ptr, col, nodes, _ = sample_adj_c(indptr, indices, batch, k, False)
        adjs.append(((row_ptr, col), batch.numel()))
        batch = nodes
    return batch, adjs[::-1]


HOP_SEED_STRIDE = 0x9E3779B97F4A7C15  # the hops of one batch draw from seeds this far apart


def sample_blocks_padded(indptr, indices, seeds, fanouts, seed=0, seed_dev=None, block32=True):
    """sample_blocks with every size fixed by (len(seeds), fanouts) -- nothing depends on what was sampled, nothing
    synchronises: the form a captured step needs (cogdl_amd.graphs.capture).  Hop h samples len(seeds_h) * fanout_h
    edge slots for its seed slots (of which the previous hop's device-side node count are in use).

    Returns (n_id, adjs, counts): n_id [cap] (unused slots hold id 0), adjs = [((row_ptr, col), n_dst_slots), ...]
    innermost block first exactly like sample_blocks -- row_ptr is padded to ALL node slots of the block, col holds
    local ids (< the number of nodes in use) with the unused slots set to 0 behind row_ptr[-1] --, and counts = the
    per-hop device tensors {nodes, edges, flags} (first hop first; the rows of ONE [hops, 3] table, `counts[0]._base`)
    for whoever wants to look (that read synchronises).
    The in-use prefix of every output equals sample_blocks' result for the same per-hop seeds.  block32: the sampler also
    writes every block in the SpMM's form (int32 + 1 / in-degree; graph_build.block_for_spmm then launches nothing)."""
    adjs, counts = [], []
    batch, count = seeds, None
    table = torch.empty((len(fanouts), 3), dtype=torch.long, device=indptr.device)
    for hop, k in enumerate(fanouts):
        hop_seed = (int(seed) + hop * HOP_SEED_STRIDE) % (1 << 64)
        row_ptr, col, nodes, _, cnt = sample_adj_padded(indptr, indices, batch, k, False, seed=hop_seed, seed_dev=seed_dev,
                                                         count=count, counts_out=table[hop], block32=block32)
        adjs.append(((row_ptr, col), batch.numel()))
        counts.append(cnt)
        batch, count = nodes, cnt[0:1]
    return batch, adjs[::-1], counts


class CapturedMiniBatchStep:
    """One sampled training step -- every sampling hop, the feature gather, forward, loss, backward, optimizer -- as ONE
    hipGraph replay (a sampled step is ~150 short launches: eager, the host's launch rate bounds it, not the GPU;
    measured on configs[3]'s shape: 1.94 -> 0.59 ms per 1024-seed step).

        indptr, indices : the graph's CSR on the GPU (int64)
        x, y            : node features (GPU, or pinned host memory: gathered zero-copy) and labels (GPU)
        forward         : forward(x_batch, blocks) -> logits for the seed slots, blocks as sample_blocks_padded returns
                          them ([((row_ptr, col), n_dst_slots), ...], innermost first).  CogDL's unchanged Graphsage
                          takes `[(None, Graph(row_ptr=rp, col=col, edge_weight=ones(len(col))), (len(rp) - 1, n))]`
        optimizer       : created with capturable=True
        initial_seeds   : [batch_size] distinct node ids; the warm-up runs of the capture are REAL training steps on them
        process_group   : replicas (configs[3]: every rank holds the graph and draws its own seeds): the gradients are
                          averaged over the group INSIDE the captured step -- one flat all-reduce (RCCL) between
                          backward and the optimizer, what torch DDP does with its buckets (cogdl/trainer/
                          trainer.py:291-303), as a node of the same graph; `params` = the tensors to average
                          (default: every parameter the optimizer holds)
        side_stream     : run what does not depend on the forward pass (the seeds' labels, the block transposes of the
                          backward) on a second stream = a second branch of the captured graph.  Default OFF: measured
                          on MI355X / ROCm 7.0 the two-branch graph replays SLOWER (1024 seeds 0.354 -> 0.439 ms, 128
                          seeds 0.272 -> 0.383 ms: a cross-branch dependency costs more than the ~30 us it hides)
    Every step: `loss = step(seeds)` copies the seeds into the static buffer and replays (the returned loss tensor is
    static too: read it when you need it, not every step).  The RNG seed lives in device memory and advances inside the
    graph.  All operators run in plan.transient_structures() mode: nothing is hashed, cached or read back."""

    def __init__(self, indptr, indices, x, y, forward, optimizer, initial_seeds, fanouts, loss_fn=None, seed=0, warmup=3,
                 process_group=None, params=None, side_stream=False):
        import torch.nn.functional as F

        if process_group is not None:
            import torch.distributed as dist

            world = dist.get_world_size(process_group)
            if params is None:
                params = [q for grp in optimizer.param_groups for q in grp["params"]]
            params = [q for q in params if q.requires_grad]

            def average_gradients():
                flat = torch.cat([q.grad.reshape(-1) for q in params])
                dist.all_reduce(flat, group=process_group)
                flat.div_(world)
                off = 0
                for q in params:
                    n = q.numel()
                    q.grad.copy_(flat[off:off + n].view_as(q.grad))
                    off += n
        else:
            def average_gradients():
                pass

        self.seeds = initial_seeds.to(device=indptr.device, dtype=torch.long).contiguous().clone()
        self.seed_dev = torch.zeros(1, dtype=torch.long, device=indptr.device)
        # side_stream: work that does not depend on the forward pass runs BESIDE it, on a second stream (a second branch
        # of the captured graph): the labels of the seeds, and the transposes the backward will need (launched by the
        # forward calls, plan.transient_structures(side_stream=...)).
        side = torch.cuda.Stream(device=indptr.device) if side_stream else None
        labels = torch.empty(self.seeds.shape, dtype=y.dtype, device=y.device)
        self._static = (labels, side)  # the captured kernels write `labels` through a raw pointer: it lives as long as the graph
        loss_fn = loss_fn or F.cross_entropy
        fanouts = list(fanouts)

        def step():
            main = torch.cuda.current_stream(indptr.device)
            with _plan.transient_structures(side_stream=side):
                if side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        torch.index_select(y, 0, self.seeds, out=labels)
                    labels_ready = torch.cuda.Event()
                    labels_ready.record(side)
                else:
                    torch.index_select(y, 0, self.seeds, out=labels)
                n_id, blocks, counts = sample_blocks_padded(indptr, indices, self.seeds, fanouts, seed=seed,
                                                            seed_dev=self.seed_dev)
                # (ids straight from the sampler; a bad one would show up in the last hop's flag word, which that hop's
                #  kernels rewrite on every replay -- no fill launch for a flag of the gather's own)
                xb = gather_rows_by_id(x, n_id, flag_word=counts[-1][2:3])
                optimizer.zero_grad(set_to_none=True)
                logits = forward(xb, blocks)
                if side is not None:
                    main.wait_event(labels_ready)  # (not the whole side stream: a transpose may still be running there)
                loss = loss_fn(logits, labels)
                loss.backward()
                if side is not None:
                    main.wait_stream(side)  # (a transpose no backward asked for must still join before the capture ends)
                average_gradients()
                optimizer.step()
                self.seed_dev.add_(1)
            return loss.detach(), counts

        self._replay = graphs.capture(step, warmup=warmup)
        self.loss, self.counts = self._replay.outputs
        self.counts_table = self.counts[0]._base  # [hops, 3] = {nodes, edges, flags} per hop: one tensor to accumulate

    def __call__(self, seeds):
        self.seeds.copy_(seeds, non_blocking=True)
        self._replay()
        return self.loss

    def check(self):
        """Raise if the last replay's sampling was invalid (synchronises): a seed or neighbour id outside the graph."""
        flags = 0
        for f in self.counts_table[:, 2].tolist():
            flags |= int(f)
        if flags:
            raise _lib.BackendError("captured step: flags %d (1 = seed id out of range, 2 = neighbour id out of range, "
                                    "4 = capacity exceeded, %d = feature gather met an id outside x)" % (flags, GATHER_BAD_ID))


class BatchPipeline:
    """Iterate over mini-batches (seeds, n_id, adjs, x_batch, y_batch) with the NEXT batch's sampling and feature
    gather running on a side stream while the caller trains on the current one.

        indptr, indices : CSR of the graph on the GPU (int64)
        x               : node features, on the GPU or in pinned host memory ([N, F])
        y               : labels on the GPU (or None)
        seed_batches    : iterable of 1-D int64 seed tensors on the GPU (distinct ids per batch)
    The caller must not synchronise the device inside its step (loss.item(), .cpu()) if it wants the overlap.

    How the overlap comes about.  Batch i+1 is prepared BEFORE batch i is handed out (so its sampling is enqueued while
    step i-1 still runs), and the side stream waits only for the event recorded when batch i+1's SEEDS were drawn --
    which happened one iteration earlier, before step i-1 was enqueued -- not for the whole main stream: waiting for
    the main stream would put sample(i+1) behind step i-1 and serialise everything.  The sampler's one host read-back
    (output sizes) waits on the side stream only.  GPU timeline: step i-1 || sample + gather i+1, step i || i+2, ...

    CONTRACT (what the overlap costs).  The side stream is NOT ordered behind the consumer's steps, so while iterating
      * `indptr`, `indices`, `x` and `y` must stay IMMUTABLE: a step that updates features, labels or the graph in place
        (feature caching, label propagation) races with the prefetch of the next batches;
      * the seed iterator is consumed TWO items ahead of the step being run (a stateful iterator -- curriculum, hard
        negatives chosen from the last step's loss -- sees the step's effect two batches late).
    `strict=True` gives both up for safety: the side stream waits for the main stream before every prefetch (sampling of
    batch i+1 then starts only after step i-1 has finished: the pre-round-3 behaviour, no overlap with the step) and
    seeds are drawn one batch ahead only."""

    def __init__(self, indptr, indices, x, y, seed_batches, fanouts, strict=False):
        self.indptr, self.indices, self.x, self.y = indptr, indices, x, y
        self.seed_batches, self.fanouts = seed_batches, list(fanouts)
        self.dev = indptr.device
        self.side = torch.cuda.Stream(device=self.dev)
        self.strict = bool(strict)

    def _draw(self, it):
        """The next seed tensor + an event on the caller's stream behind whatever produced it."""
        try:
            seeds = next(it)
        except StopIteration:
            return None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.dev))
        return seeds, ready

    def _prepare(self, drawn):
        seeds, ready = drawn
        if self.strict:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)  # the seeds exist; nothing else of the caller's stream is waited for
            if seeds.is_cuda:
                seeds.record_stream(self.side)  # allocated on the caller's stream, read on this one
            n_id, adjs = sample_blocks(self.indptr, self.indices, seeds, self.fanouts)
            xb = gather_rows_by_id(self.x, n_id)
            yb = None if self.y is None else self.y.index_select(0, seeds)
            done = torch.cuda.Event()
            done.record(self.side)
        return seeds, n_id, adjs, xb, yb, done

    def __iter__(self):
        it = iter(self.seed_batches)
        drawn = self._draw(it)
        if drawn is None:
            return
        cur = self._prepare(drawn)
        drawn = self._draw(it)            # seeds of batch 1, drawn before step 0 is enqueued
        while cur is not None:
            nxt_drawn = self._draw(it) if drawn is not None else None  # seeds of batch i+2: before step i is enqueued
            nxt = self._prepare(drawn) if drawn is not None else None  # batch i+1: overlaps with step i-1 (running)
            seeds, n_id, adjs, xb, yb, done = cur
            main = torch.cuda.current_stream(self.dev)
            main.wait_event(done)
            for t in [seeds, n_id, xb] + ([yb] if yb is not None else []) + [b for (blk, _) in adjs for b in blk]:
                t.record_stream(main)  # allocated on the side stream, consumed on the caller's
            yield seeds, n_id, adjs, xb, yb  # the consumer enqueues step i (asynchronously) after this
            cur, drawn = nxt, nxt_drawn


def layerwise_inference(convs, x_all, indptr, indices, batch_size=65536, activation=torch.relu, make_graph=None,
                        keep_on_host=None):
    """Graphsage.inference (cogdl/models/nn/graphsage.py:106-119) for a GPU-resident graph: for every layer, for every
    block of `batch_size` target nodes, the FULL neighbourhood is taken with sample_adj(-1) on the GPU, the current
    layer's inputs of the block's frontier are gathered with gather_rows_by_id, and `convs[i](make_graph(row_ptr, col),
    x)[:block]` is evaluated; the activation follows every layer but the last.

    x_all may live on the GPU or in pinned host memory; intermediate layer outputs follow `keep_on_host` (default: the
    same side as x_all) -- on the host they are written into a pinned buffer, so the next layer reads them zero-copy.
    `make_graph(row_ptr, col)` builds what the layer takes as its graph argument (default: the tuple; with CogDL:
    `lambda rp, c: Graph(row_ptr=rp, col=c)`)."""
    dev = indptr.device
    n = indptr.numel() - 1
    if keep_on_host is None:
        keep_on_host = not x_all.is_cuda
    make_graph = make_graph or (lambda rp, c: (rp, c))
    with torch.no_grad():
        for li, conv in enumerate(convs):
            out_all = None
            for start in range(0, n, batch_size):
                batch = torch.arange(start, min(n, start + batch_size), device=dev)
                row_ptr, col, nodes, _ = sample_adj_c(indptr, indices, batch, -1, False)
                x = gather_rows_by_id(x_all, nodes)
                h = conv(make_graph(row_ptr, col), x)[: batch.numel()]
                if li != len(convs) - 1 and activation is not None:
                    h = activation(h)
                if out_all is None:
                    shape = (n,) + tuple(h.shape[1:])
                    out_all = (torch.empty(shape, dtype=h.dtype, pin_memory=True) if keep_on_host
                               else torch.empty(shape, dtype=h.dtype, device=dev))
                out_all[start:start + batch.numel()].copy_(h, non_blocking=True)
            if keep_on_host:
                torch.cuda.current_stream(dev).synchronize()  # the pinned buffer is complete before it is read again
            x_all = out_all
    return x_all
