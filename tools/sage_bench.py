#!/usr/bin/env python3
"""BASELINE.json configs[3] end to end (not bench.py's contract line; a profile for SURVEY.md section 8e/8f):
GraphSAGE on an ogbn-products-shaped graph with neighbour-sampling mini-batches, everything resident in HBM.

One training step = what cogdl/data/sampler.py:NeighborSampler.sample + cogdl/models/nn/graphsage.py:forward do per
mini-batch -- sample_adj per layer (fan-out [10, 10], without replacement), gather the input features of the outermost
frontier, two SAGELayers (mean aggregator = row-normalised csr_spmm over the sampled block, cat, Linear; relu +
dropout in between), cross-entropy, Adam -- with the graph (118 M edges), the features ([N, 100] fp32 = 0.98 GB) and
the sampler on the GPU: sample_adj_c -> cogdl_hip_sample_adj, aggregation -> csrspmm -> cogdl_hip_csr_spmm.
The reference samples on the CPU in DataLoader workers (sample.cpp, single-threaded per worker), gathers x[n_id] on the
host and copies every batch over PCIe.

N > 1 GPUs (python -m torch.distributed.run --nproc-per-node N tools/sage_bench.py): independent replicas, the
gradients all-reduced by torch DDP over RCCL as in cogdl/trainer/trainer.py:291-303.  Prints one JSON line.
Usage: python tools/sage_bench.py [--batch 1024] [--steps 50] [--nodes 2449029] [--degree 50.5]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogdl_amd import synth, transient_structures  # noqa: E402
from cogdl_amd.graph_build import block_for_spmm  # noqa: E402
from cogdl_amd.operators.spmm import csrspmm, csrspmm_block  # noqa: E402
from cogdl_amd.pipeline import (BatchPipeline, CapturedMiniBatchStep, gather_rows_by_id,  # noqa: E402
                                layerwise_inference, sample_blocks)


class SageMean(torch.nn.Module):
    """SAGELayer(aggr='mean') of cogdl/layers/sage_layer.py:8-12,69-87: fc(cat(x, row_norm(A) x))."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc = torch.nn.Linear(2 * in_feats, out_feats)

    def forward(self, block, x):
        row_ptr, col = block
        deg = (row_ptr[1:] - row_ptr[:-1])
        w = torch.repeat_interleave(1.0 / deg.clamp(min=1).float(), deg)  # Graph.row_norm(): 1 / in-degree per edge
        h = csrspmm(row_ptr.int(), col.int(), x, w)  # the .int() copies CogDL's dispatcher makes (spmm_utils.py:106)
        return self.fc(torch.cat([x, h], dim=-1))

    def forward_padded(self, block, n_dst, x):
        """The same layer on a fixed-capacity block (sample_blocks_padded), for the n_dst target slots only (the
        caller keeps just those rows anyway, graphsage.py:99): the mean is the SpMM's in_norm epilogue, the block's
        transpose for the backward is taken on the spot (csrspmm_block) -- every shape is static, nothing syncs."""
        rp, col, inv_deg = block_for_spmm(block[0], block[1], n_dst)  # int32 indices + 1 / in-degree: one launch
        h = csrspmm_block(rp, col, x, None, inv_deg)
        return self.fc(torch.cat([x[:n_dst], h], dim=-1))


class Sage(torch.nn.Module):
    def __init__(self, feats, hidden, classes):
        super().__init__()
        self.convs = torch.nn.ModuleList([SageMean(feats, hidden), SageMean(hidden, classes)])

    def forward(self, x, adjs):  # graphsage.py:93-104
        for i, (block, n_dst) in enumerate(adjs):
            x = self.convs[i](block, x)[:n_dst]
            if i != len(adjs) - 1:
                x = F.dropout(F.relu(x), p=0.5, training=self.training)
        return x

    def forward_padded(self, x, adjs):
        for i, (block, n_dst) in enumerate(adjs):
            x = self.convs[i].forward_padded(block, n_dst, x)
            if i != len(adjs) - 1:
                x = F.dropout(F.relu(x), p=0.5, training=self.training)
        return x


def captured_training(args, dev, n, indptr, indices, x_all, y_all, model, gen, group=None):
    """--captured: the whole mini-batch step -- both sampling hops, the feature gather, forward, backward, Adam -- as ONE
    hipGraph launch per step (cogdl_amd.pipeline.CapturedMiniBatchStep).  Every buffer has the capacity
    B * (1 + 10) * (1 + 10) node slots; the seeds of a step are copied into a static buffer before the replay."""
    b = args.batch
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True, fused=True)  # one kernel per step, not ~10
    order = torch.randperm(n, device=dev, generator=gen)  # distinct seeds per batch, as a DataLoader over the train set gives
    model.train()
    step = CapturedMiniBatchStep(indptr, indices, x_all, y_all, model.forward_padded, opt, order[:b], [10, 10], seed=20240 + int(os.environ.get("RANK", 0)),
                                 process_group=group, side_stream=args.side_stream)  # replicas: the gradient all-reduce is a node of the captured graph
    n_batches = n // b
    for i in range(args.warmup):
        step(order[(i % n_batches) * b:(i % n_batches + 1) * b])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = torch.zeros_like(step.counts_table)  # {nodes, edges, flags} per hop, summed over the steps: one launch per step
    for i in range(args.steps):
        k = (args.warmup + i) % n_batches
        loss = step(order[k * b:(k + 1) * b])
        acc += step.counts_table
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step.check()
    if not bool(torch.isfinite(loss)):
        raise SystemExit("captured step: loss %s" % float(loss))
    if int(acc[:, 2].sum()):
        raise SystemExit("captured step: a replay raised flags (per-hop sums %s)" % acc[:, 2].tolist())
    return dt, b * args.steps, int(acc[-1, 0]), int(acc[:, 1].sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=2_449_029)
    ap.add_argument("--degree", type=float, default=50.5)
    ap.add_argument("--feat", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--features", default="hbm", choices=["hbm", "host"],
                    help="host: the feature matrix stays in PINNED host memory and x[n_id] is gathered zero-copy over the host link")
    ap.add_argument("--pipeline", action="store_true", help="sample + gather batch i+1 on a side stream while batch i trains")
    ap.add_argument("--captured", action="store_true",
                    help="fixed-capacity blocks, the whole step (sampling included) replayed as one hipGraph (single GPU)")
    ap.add_argument("--side-stream", action="store_true",
                    help="--captured: labels and block transposes on a second branch of the captured graph (A/B; slower on ROCm 7.0)")
    ap.add_argument("--inference", action="store_true", help="also time layer-wise full-neighbour inference over all nodes")
    ap.add_argument("--torch-linear", action="store_true",
                    help="keep torch / hipBLASLt for the SAGE layers' Linear (default: cogdl_amd.linear = install(linear=True): "
                         "the weight gradient of the first layer is a reduction over all ~1e4..1e5 frontier rows that hipBLASLt "
                         "runs on 28 workgroups, 72 us of a 0.44 ms captured step)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if not args.torch_linear:
        from cogdl_amd import linear as cogdl_linear

        cogdl_linear.install()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n = args.nodes
    src, dst = synth.rmat_pairs(n, int(n * args.degree / 2), 0, device=dev)  # every replica holds the whole graph
    g = synth.finalize(src, dst, n, norm=None, self_loops=False)
    del src, dst
    indptr, indices = g.rowptr.long(), g.colind.long()
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    x_all = torch.randn(n, args.feat, device=dev, generator=gen)
    if args.features == "host":
        x_all = x_all.cpu().pin_memory()
    y_all = torch.randint(0, args.classes, (n,), device=dev, generator=gen)
    torch.manual_seed(0)
    model = Sage(args.feat, args.hidden, args.classes).to(dev)
    # (--captured averages the gradients itself, inside the captured graph: no DDP hooks on the parameters then; the
    #  replicas start from the same weights because every rank seeds the initialisation alike)
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index]) if world > 1 and not args.captured else model
    opt = torch.optim.Adam(net.parameters(), lr=0.01)

    def draw_seeds():
        return torch.randint(0, n, (args.batch,), device=dev, generator=gen).unique()

    def train_on(seeds, n_id, adjs, xb, yb):
        opt.zero_grad(set_to_none=True)
        with transient_structures():  # sampled blocks never repeat: transposed on the spot, not hashed into the plan cache
            loss = F.cross_entropy(net(xb, adjs), yb)
            loss.backward()
        opt.step()
        return seeds.numel(), n_id.numel(), sum(b[0][1].numel() for b in adjs)

    def step():
        seeds = draw_seeds()
        n_id, adjs = sample_blocks(indptr, indices, seeds, [10, 10])
        return train_on(seeds, n_id, adjs, gather_rows_by_id(x_all, n_id), y_all[seeds])

    net.train()
    if not args.captured:
        for _ in range(args.warmup):
            step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seeds = nodes = edges = 0
    if args.captured:
        secs, seeds, nodes, edges = captured_training(args, dev, n, indptr, indices, x_all, y_all, model, gen,
                                                      torch.distributed.group.WORLD if world > 1 else None)
        t0 = time.perf_counter() - secs
    elif args.pipeline:
        batches = [draw_seeds() for _ in range(args.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for batch in BatchPipeline(indptr, indices, x_all, y_all, batches, [10, 10]):
            a, b, c = train_on(*batch)
            seeds, nodes, edges = seeds + a, nodes + b, edges + c
    else:
        for _ in range(args.steps):
            a, b, c = step()
            seeds, nodes, edges = seeds + a, nodes + b, edges + c
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    tot = torch.tensor([seeds], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tot)
    # the sampling part alone (its own loop: no synchronisation inside the timed training steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        sample_blocks(indptr, indices, torch.randint(0, n, (args.batch,), device=dev, generator=gen).unique(), [10, 10])
    torch.cuda.synchronize()
    ms_sample = (time.perf_counter() - t1) / 20 * 1e3
    # the feature gather alone, for a typical frontier
    n_id, _ = sample_blocks(indptr, indices, draw_seeds(), [10, 10])
    for _ in range(3):
        gather_rows_by_id(x_all, n_id)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        gather_rows_by_id(x_all, n_id)
    torch.cuda.synchronize()
    ms_gather = (time.perf_counter() - t1) / 20 * 1e3
    gather_gbs = n_id.numel() * args.feat * 4 / ms_gather / 1e6
    infer = None
    if args.inference and rank == 0:
        model.eval()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = layerwise_inference(list(model.convs), x_all, indptr, indices, batch_size=65536)
        torch.cuda.synchronize()
        infer = {"s": time.perf_counter() - t1, "nodes": n, "out_shape": list(out.shape),
                 "what": "Graphsage.inference: 2 layers x all nodes, full neighbourhoods (sample_adj(-1) on the GPU)"}
    if rank == 0:
        print(json.dumps({
            "metric": "GraphSAGE mini-batch training, seed nodes/s (products-shaped graph, fan-out [10,10]) @%d GPU" % world,
            "value": float(tot) / float(dt), "unit": "seed nodes/s", "n_gpus": world, "steps": args.steps,
            "ms_per_step": float(dt) / args.steps * 1e3, "ms_sampling_alone_rank0": ms_sample,
            "ms_feature_gather_alone_rank0": ms_gather, "feature_gather_GBs": gather_gbs, "layerwise_inference": infer,
            "batch": args.batch, "frontier_nodes_per_step": nodes // args.steps, "sampled_edges_per_step": edges // args.steps,
            "config": {"nodes": n, "nnz": int(g.nnz), "feat": args.feat, "hidden": args.hidden, "classes": args.classes,
                       "linear": "torch / hipBLASLt" if args.torch_linear else "cogdl_amd.linear (MFMA weight gradient)",
                       "sampler": "cogdl_hip_sample_adj (GPU-resident graph)", "features": "resident in HBM" if args.features == "hbm" else "pinned host memory, zero-copy gather",
                       "pipeline": ("whole step (2 sampling hops, gather, fwd, bwd, Adam) = one hipGraph replay over fixed-capacity blocks" if args.captured
                                    else "next batch sampled + gathered on a side stream" if args.pipeline else "none"),
                       "parallelism": "replicas + DDP all-reduce (RCCL)" if world > 1 else "single GPU"}}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
