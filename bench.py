#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the sparse message-passing hot path.

N = 1 : BASELINE.json configs[1] -- "GCN on ogbn-arxiv (170k nodes, 1.2M edges, 128-dim feat), full-graph
        csr_spmm fwd+bwd on 1 MI355X, fp32".  A step = one csrspmm forward + its backward (grad wrt the
        features) through the autograd operator a CogDL layer would call, on the arxiv-shaped synthetic graph
        (uniform-random topology = worst-case locality; no dataset is available offline), inputs resident in HBM.
        value = GEdges/s = 2 * nnz * steps / time.
N > 1 : configs[4] -- vertex-sharded csr_spmm (1-D row partition, halo rows exchanged with an RCCL all-to-all
        overlapped with the local-column SpMM) over THE graph the N = 1 line's configs4_papers_1gpu leg runs: the
        papers100M-shaped symmetrised graph (3.2e9 edges), every rank keeping the rows of its edge-balanced contiguous
        range (cogdl_amd/dist.py: papers_graph_shard) -- strong scaling; halo rows and remote edge share are MEASURED on
        that partition and reported, value_at_world1 repeats the one-GPU figure of the same metric.
        value = global nnz * 2 * steps / time (max over ranks).  Started without a launcher, `bench.py --gpus N` spawns
        its N ranks itself (torch.distributed.run on 127.0.0.1); a rank count that differs from --gpus is refused.
        Follow-up legs run in child process groups with hard timeouts and land in the same line: assumed_partition (the
        generated shards of rounds 2-5: one papers100M/8-sized shard per GPU, weak scaling, --remote-frac / --halo-frac
        as INPUTS = what a locality-preserving partition is assumed to leave, with its `predicted` model),
        worst_case_partition (random partition: (N-1)/N of the sources remote) and configs3_sage_replicas
        (tools/sage_bench.py --captured).

One JSON line on stdout (rank 0).  Extra objects: roofline (dominant kernel, HIP-event timed inside the timed region;
`traffic` = PMC bytes measured by rocprofv3 passes inside this run; `rmat` and `hbm_resident` = the same kernel on the
power-law topology and on a shard far beyond the caches) and cpu_baseline (the reference's own csr_spmm_cpu, built from
/root/reference by oracle/Makefile, timed on this host's cores; N=1 only).  The N = 1 line also carries configs3_sage: the
captured GraphSAGE mini-batch step of configs[3] on this GPU (child interpreter), the base of the replica legs, and
configs2_gat: the GAT training step of configs[2] (Reddit-shaped graph, bf16) with the model's default arguments through
the fused attention-dropout operator, beside the unchanged layer's time and the per-kernel roofline fractions;
configs4_papers_1gpu (round 5): BASELINE configs[4] at FULL size on this one GPU -- the papers100M-shaped graph, 1.6e9 directed
and 3.2e9 symmetrised edges, F = 128 fp32, through csrspmm with 64-bit row pointers (tools/papers_bench.py): the N = 1 end of
the 1 -> 8 curve; and roofline.measured_copy_GBs / measured_read_GBs: the box's own roofs, measured in the same run
(fractions of the 8 TB/s spec number are not comparable across boxes).  Round 6: configs4_papers_1gpu.<graph>.traffic = the HBM-side
bytes of a full forward pass per graph (rocprofv3 --pmc inside this run), gnn_epoch.accounting = library / torch kernel time and
launch gaps of the captured GCN epoch (rocprofv3 --kernel-trace inside this run), configs2_gat.roofline.*.l2_frac = the fused GAT
kernels against the aggregate L2 bandwidth (their tables are cache-resident).
"""
import argparse
import json
import os
import sys
import time

T0 = time.time()  # before `import torch` (minutes on a fresh box): the N>1 legs are budgeted against the whole command

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves


def b_alg(nnz, m, f, s=4, s_w=4):
    """Algorithmic bytes of one csr_spmm launch: per edge colind + weight + one gathered row,
    per row rowptr + one output row."""
    return nnz * (4 + s_w + f * s) + m * (4 + f * s)


def cpu_baseline(g, x, budget_s=10.0):
    """Reference csr_spmm_cpu on the host cores (oracle/_ref, -O3 build; as-shipped -O0 build timed once)."""
    from oracle import oracle

    host_cores = os.cpu_count() or 1
    res = {"unit": "GEdges/s", "host_cores": host_cores}
    if oracle.ref_available("O3"):
        fn = oracle.ref_spmm_cpu("O3")
        res["kind"] = "reference"

        def run(nt):
            torch.set_num_threads(nt)  # = omp_set_num_threads: the reference's `#pragma omp parallel for`
            return fn(g.rowptr, g.colind, g.weight, x)
        flavour = "cogdl/operators/spmm/spmm_cpu.cpp built -fopenmp -O3 -mavx2 -mfma"
    else:
        res["kind"] = "port"

        def run(nt):
            return oracle.csr_spmm(g.rowptr, g.colind, g.weight, x, nthreads=nt)
        flavour = "oracle/cogdl_oracle.c (OpenMP port)"
    # The reference's dynamic-schedule loop does not scale to every core of a big host (256 threads were 12x
    # slower than 8 on the GPU box): probe a few thread counts and report the BEST one, `cores` = threads used.
    probe = {}
    for nt in sorted({n for n in (4, 8, 16, 32, 64, 128, host_cores) if n <= host_cores}):
        run(nt)
        t0 = time.perf_counter()
        run(nt)
        probe[nt] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    reps = int(max(3, min(400, budget_s / max(probe[cores], 1e-4))))
    t0 = time.perf_counter()
    for _ in range(reps):
        run(cores)
    dt = (time.perf_counter() - t0) / reps
    res["cores"] = cores
    res["value"] = g.nnz / dt / 1e9
    res["ms_per_call"] = dt * 1e3
    res["ms_per_call_by_threads"] = {str(k): round(v * 1e3, 2) for k, v in probe.items()}
    res["sample"] = "%d forward csr_spmm calls on the full arxiv-like graph (nnz=%d, F=%d), %s, %d threads" % (
        reps, g.nnz, x.shape[1], flavour, cores)
    torch.set_num_threads(host_cores)
    if oracle.ref_available("asshipped"):
        fn0 = oracle.ref_spmm_cpu("asshipped")
        fn0(g.rowptr, g.colind, g.weight, x)
        t0 = time.perf_counter()
        for _ in range(3):
            fn0(g.rowptr, g.colind, g.weight, x)
        res["as_shipped_O0_GEdges_s"] = g.nnz / ((time.perf_counter() - t0) / 3) / 1e9
    return res


def pmc_traffic(topology, feat, live=True):
    """`roofline.traffic`: HBM-side bytes per csr_spmm launch from the PMC counters.  Measured IN THIS RUN
    (tools/pmc_live.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over a probe with the same graph and
    kernel, calibrated on a 1 GiB copy as the guide prescribes for gfx950); only when rocprofv3 is unavailable or fails
    does the line fall back to the committed summary of tools/gpu_round.sh pmc -- and says so.  -> (bytes | None, info)"""
    if live:
        from tools import pmc_live

        r = pmc_live.collect(topology, feat)
        if "hbm_bytes_per_launch" in r:
            return r["hbm_bytes_per_launch"], r
        info = {"live_error": r.get("error")}
    else:
        info = {"live_error": "skipped (--no-pmc)"}
    v, path = load_pmc_traffic("arxiv_%s_F%d" % (topology, feat))
    info["source"] = "COMMITTED profile %s (not measured in this run)" % path if v is not None else None
    return v, info


def load_pmc_traffic(phase="arxiv_uniform_F128"):
    """HBM-side bytes per launch of the main csr_spmm kernel from the committed rocprofv3 --pmc summary
    (profiles/rNN_pmc_spmm_arxiv.json, the latest round's; made by tools/gpu_round.sh pmc + tools/pmc_summarize.py: separate passes for
    FETCH_SIZE and WRITE_SIZE, FETCH_SIZE calibrated on a 1 GiB copy = the guide's gfx950 x2 correction).
    None when no summary is committed (PMC counters cannot be read from inside the bench process)."""
    import glob

    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_spmm_arxiv.json")))[-1]
        return json.load(open(path))["phases"][phase]["main"]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def gcn_epoch_ms(gd, rowptr64, colind64, x, reps=20, warmup=5, mfma_linear=True, captured=False, return_step=False):
    """The 'GNN epoch time' half of BASELINE.json's metric: one full-graph training step (= one epoch) of CogDL's
    default `gcn` model (cogdl/models/nn/gcn.py:26-29: 2 GCNLayers, hidden 64, relu, dropout 0.5; GCNLayer.forward
    = spmm(graph, linear(x)), cogdl/layers/gcn_layer.py:51-53) on the arxiv-shaped graph, 40 classes, Adam(lr 0.01,
    wd 5e-4) -- the aggregation goes through the csrspmm operator exactly as spmm_utils.spmm calls it (fresh .int()
    index copies per call), the dense X.W stays on torch/hipBLASLt.  Median of `reps` steps, fenced."""
    from cogdl_amd import linear as cogdl_linear
    from cogdl_amd.operators.spmm import csrspmm

    if mfma_linear:  # unchanged torch.nn.Linear modules pick the MFMA weight-gradient kernel up through F.linear
        cogdl_linear.install()
    dev = x.device
    torch.manual_seed(0)
    n, f_in, hidden, classes = x.shape[0], x.shape[1], 64, 40
    lin1, lin2 = torch.nn.Linear(f_in, hidden).to(dev), torch.nn.Linear(hidden, classes).to(dev)
    drop = torch.nn.Dropout(0.5)
    params = list(lin1.parameters()) + list(lin2.parameters())
    # (captured leg: torch's single-kernel Adam -- the same update, one graph node instead of eight)
    opt = torch.optim.Adam(params, lr=0.01, weight_decay=5e-4, capturable=captured, fused=True if captured else None)
    y = torch.randint(0, classes, (n,), device=dev)
    train_mask = torch.rand(n, device=dev) < 0.537  # ogbn-arxiv: 90,941 of 169,343 nodes train
    train_idx = torch.nonzero(train_mask).flatten()  # captured variant: static shapes, no boolean-mask indexing
    y_train = y[train_idx]
    feats = x.detach()

    def step():
        opt.zero_grad(set_to_none=True)
        h = csrspmm(rowptr64.int(), colind64.int(), lin1(feats), gd.weight, True)
        h = drop(torch.relu_(h))
        out = csrspmm(rowptr64.int(), colind64.int(), lin2(h), gd.weight, True)
        if captured:
            loss = torch.nn.functional.cross_entropy(out.index_select(0, train_idx), y_train)
        else:
            loss = torch.nn.functional.cross_entropy(out[train_mask], y[train_mask])
        loss.backward()
        opt.step()

    if captured:  # the whole step (forward, backward, Adam) as ONE hipGraph launch (cogdl_amd/graphs.py)
        from cogdl_amd import graphs

        step = graphs.capture(step, warmup=3)
    for _ in range(warmup):
        step()
    if return_step:  # (tools/epoch_account.py replays the step under rocprofv3 --kernel-trace; the MFMA linear hook stays installed)
        return step
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    cogdl_linear.uninstall()
    return {"model": "CogDL gcn default: 2 x GCNLayer, hidden 64, relu, dropout 0.5, Adam; full-graph step = 1 epoch",
            "ms": ts[len(ts) // 2], "min_ms": ts[0], "reps": reps,
            "spmm_calls_per_epoch": 4, "spmm_widths": [hidden, classes],
            "linear": "cogdl_amd.linear: MFMA forward / grad_input / split-K weight gradient" if mfma_linear else "torch / hipBLASLt"}


def trainer_epoch(budget_s=240):
    """`gnn_epoch.trainer_ms`: the reference's OWN Trainer.train_step (cogdl/trainer/trainer.py:500-540) timed inside
    `experiment(model='gcn', dataset=<arxiv-shaped NodeDataset>)`, run by the unchanged reference package (staged
    copy under oracle/_ref/pkg) in child interpreters -- on cuda:0 on top of cogdl_amd.install() (with and without the
    MFMA linear hook) and on the reference's own CPU path beside it.  None where the staged package is absent."""
    import subprocess

    from tools import refpkg

    if not refpkg.available():
        return None
    out = {}
    # arxiv_example_*: the model of the reference's ogbn-arxiv example (3 GCNLayers, hidden 256, batch-norm; examples/ogb/arxiv/
    # gnn.py:148-156), the configuration BASELINE.md section 3 timed at 5.94 s per training step on the CPU
    for key, argv in (("gpu", ["gpu", "30"]), ("gpu_mfma_linear", ["gpu", "30", "linear"]),
                      ("gpu_mfma_linear_structure_memo", ["gpu", "30", "linear", "memo"]), ("cpu_reference", ["cpu", "3"]),
                      ("gpu_fp16", ["gpu", "30", "fp16"]), ("gpu_fp16_mfma_linear", ["gpu", "30", "linear", "fp16"]),
                      ("arxiv_example_gpu", ["gpu", "30", "example"]),
                      ("arxiv_example_gpu_mfma_linear", ["gpu", "30", "linear", "example"])):
        try:
            proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trainer_epoch.py")] + argv,
                                  capture_output=True, text=True, timeout=budget_s)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith("TRAINER ")]
            out[key] = json.loads(line[-1][8:]) if line else {"error": (proc.stderr or proc.stdout)[-400:]}
        except subprocess.TimeoutExpired:
            out[key] = {"error": "timeout after %d s" % budget_s}
    return out


def kernel_alone_ms(gd, xs, reps=50):
    """The csr_spmm kernel alone (no autograd, no per-call index casts), HIP-event timed on the launch stream."""
    from cogdl_amd.operators import spmm as spmm_mod

    with torch.no_grad():
        for _ in range(5):
            spmm_mod.csr_spmm_raw(gd.rowptr, gd.colind, gd.weight, xs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            spmm_mod.csr_spmm_raw(gd.rowptr, gd.colind, gd.weight, xs)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rmat_roofline(dev, feats=(128, 64, 40)):
    """`roofline.rmat`: the same kernel on the REALISTIC topology -- the arxiv-sized R-MAT graph (power-law degrees,
    hub rows of 1e4 edges: the real ogbn-arxiv is power-law, the headline's uniform graph never triggers the long-row
    path) at the headline width and at the two widths CogDL's gcn runs (hidden 64, 40 classes)."""
    from cogdl_amd import synth

    from cogdl_amd import _lib, xcdplan
    from cogdl_amd.operators import spmm as spmm_mod

    g = synth.arxiv_like(seed=0, topology="rmat")
    gd = g.to(dev)
    out = {"topology": "arxiv-sized R-MAT (a=.57,b=.19,c=.19), nnz=%d, max degree %d" % (g.nnz, int(g.degrees().max())),
           "what": "kernel_ms / frac: the launch a skewed structure takes once its fingerprint is on the host (backward "
                   "passes from the second sighting on; forward calls under install(structure_memo=True) or with the same index tensors) -- cogdl_hip_csr_spmm_xcd over a plan cut at the "
                   "exact-row bound, virtual rows in order of length (cogdl_amd/xcdplan.py: ordered_wanted); "
                   "ordinary_kernel_ms / frac_ordinary: cogdl_hip_csr_spmm (a forward call whose structure hash is still in "
                   "flight).  Rows up to the exact-row bound are bit-identical in both.  Fractions above 1: the 27-87 MB "
                   "operand is served by L2 / Infinity Cache, as on the headline."}
    plan = xcdplan.build(gd.rowptr, gd.colind, split=int(_lib.hip().cogdl_hip_exact_row_edges(g.nnz)))

    def planned_ms(xs, reps=50):
        with torch.no_grad():
            for _ in range(5):
                spmm_mod.csr_spmm_xcd_raw(plan, gd.weight, xs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                spmm_mod.csr_spmm_xcd_raw(plan, gd.weight, xs)
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for f in feats:
        xs = torch.randn(g.num_nodes, f, device=dev)
        ms0 = kernel_alone_ms(gd, xs)
        ms = planned_ms(xs)
        ach0, ach = (b_alg(g.nnz, g.num_nodes, f) / (t * 1e-3) / 1e9 for t in (ms0, ms))
        out["F%d" % f] = {"kernel_ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                          "ordinary_kernel_ms": ms0, "achieved_ordinary": ach0, "frac_ordinary": ach0 / HBM_PEAK_GBS}
    out["achieved"], out["frac"] = out["F%d" % feats[0]]["achieved"], out["F%d" % feats[0]]["frac"]
    out["frac_ordinary"] = out["F%d" % feats[0]]["frac_ordinary"]
    return out


def bench_single(args):
    from cogdl_amd import synth
    from cogdl_amd.operators import spmm as spmm_mod
    from cogdl_amd.operators.spmm import csrspmm

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    f = args.feat
    g = synth.arxiv_like(seed=0, topology=args.topology)
    x_cpu = torch.randn(g.num_nodes, f, generator=torch.Generator().manual_seed(0))
    gd = g.to(dev)
    rowptr64, colind64 = gd.rowptr.long(), gd.colind.long()  # CogDL's Graph keeps int64 indices
    x = x_cpu.to(dev).requires_grad_()
    gout = torch.randn(g.num_nodes, f, device=dev)

    def step():
        # fresh int32 copies of the structure every call, as CogDL's dispatcher does (spmm_utils.py:106)
        out = csrspmm(rowptr64.int(), colind64.int(), x, gd.weight, True)
        x.grad = None
        out.backward(gout)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events around the csr_spmm launches of every 4th step (both launches of it; at least 5 steps): a timing-event record
    # costs the host ~18 us, four per step would make the timed loop host-bound on a slow box (spmm.KernelEventLog)
    spmm_mod.KERNEL_EVENTS = spmm_mod.KernelEventLog(every=1)
    sample_every = max(1, min(4, args.steps // 5))
    t0 = time.perf_counter()
    for i in range(args.steps):
        spmm_mod.KERNEL_EVENTS.enabled = i % sample_every == 0  # (this step's two launches, or none)
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    events, spmm_mod.KERNEL_EVENTS = list(spmm_mod.KERNEL_EVENTS), None
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)

    fwd_ms = kernel_alone_ms(gd, x.detach())
    traffic, traffic_info = pmc_traffic(args.topology, f, live=not args.no_pmc)

    bytes_alg = b_alg(g.nnz, g.num_nodes, f)
    achieved = bytes_alg / (kern_ms * 1e-3) / 1e9
    result = {
        "metric": "SpMM GEdges/s (csr_spmm fwd+bwd, ogbn-arxiv-shaped GCN aggregation) @1 GPU",
        "value": 2 * g.nnz * args.steps / dt / 1e9,
        "unit": "GEdges/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ogbn-arxiv-like full-graph csr_spmm fwd+bwd (configs[1])", "nodes": g.num_nodes,
                   "nnz": g.nnz, "feat": f, "topology": args.topology, "weighted": True,
                   "parallelism": "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "rowreduce_main_kernel<SpmmOp<float,VEC=2,LPR=64,UNROLL=8,weighted,exact,no-epilogue>> (the HIP events bracket one "
                               "cogdl_hip_csr_spmm call = this kernel + the ~4 us rowreduce_combine_kernel)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_info": traffic_info,
                     "algorithmic_bytes_per_launch": bytes_alg,
                     "compulsory_bytes_per_launch": g.nnz * 8 + 4 * (g.num_nodes + 1) + 2 * g.num_nodes * f * 4,
                     "kernel_ms_in_step": kern_ms, "kernel_launches_timed": len(events),
                     "kernel_timing": "HIP events around both csr_spmm launches of every %d%s step of the timed region (a timing-event "
                                      "record costs the host ~18 us: bracketing all %d launches would make the loop host-bound on a "
                                      "slow box)" % (sample_every, {1: "st", 2: "nd", 3: "rd"}.get(sample_every, "th"), 2 * args.steps),
                     "kernel_ms_fwd_alone": fwd_ms,
                     "GEdges_s_fwd_alone": g.nnz / (fwd_ms * 1e-3) / 1e9},
    }
    result["roofline"]["rmat"] = rmat_roofline(dev)
    result["gnn_epoch"] = gcn_epoch_ms(gd, rowptr64, colind64, x)
    result["gnn_epoch"]["ms_with_torch_linear"] = gcn_epoch_ms(gd, rowptr64, colind64, x, mfma_linear=False)["ms"]
    try:
        cap = gcn_epoch_ms(gd, rowptr64, colind64, x, captured=True)
        result["gnn_epoch"]["ms_hipgraph"] = cap["ms"]
        result["gnn_epoch"]["hipgraph"] = ("the same step captured once with cogdl_amd.graphs.capture (index tensor instead of "
                                           "the boolean mask, capturable Adam) and replayed as one graph launch")
    except Exception as e:  # a capture failure must not take the bench line down
        result["gnn_epoch"]["ms_hipgraph"] = None
        result["gnn_epoch"]["hipgraph_error"] = repr(e)[:300]
    try:
        # the same epoch on the REALISTIC topology (arxiv-sized R-MAT graph: the real ogbn-arxiv is power-law): eager, and
        # captured -- where every csr_spmm launch runs over the length-ordered plan of the structure (the recorded run of
        # graphs.capture waits for the key: plan.taped_choice) -- with the plans switched off beside it
        from cogdl_amd import xcdplan

        gr = synth.arxiv_like(seed=0, topology="rmat").to(dev)
        r64r, c64r = gr.rowptr.long(), gr.colind.long()
        rm = {"topology": "arxiv-sized R-MAT (roofline.rmat's graph)", "ms": gcn_epoch_ms(gr, r64r, c64r, x)["ms"],
              "ms_hipgraph": gcn_epoch_ms(gr, r64r, c64r, x, captured=True)["ms"]}
        mode, xcdplan.MODE = xcdplan.MODE, "off"
        try:
            rm["ms_hipgraph_plans_off"] = gcn_epoch_ms(gr, r64r, c64r, x, captured=True)["ms"]
        finally:
            xcdplan.MODE = mode
        result["gnn_epoch"]["rmat"] = rm
        del gr, r64r, c64r
    except Exception as e:
        result["gnn_epoch"]["rmat"] = {"error": repr(e)[:300]}
    if not args.no_pmc:
        # where the captured epoch goes: the library's kernels / torch's kernels (three largest named) / launch gaps, from one
        # rocprofv3 --kernel-trace of the hipGraph replay in a child interpreter (tools/epoch_account.py)
        from tools import epoch_account

        result["gnn_epoch"]["accounting"] = epoch_account.collect()
    if not args.no_trainer:
        tr = trainer_epoch()
        if tr is not None:
            result["gnn_epoch"]["trainer"] = tr
            if "train_step_ms_median" in tr.get("gpu", {}):
                result["gnn_epoch"]["trainer_ms"] = tr["gpu"]["train_step_ms_median"]
            if "train_step_ms_median" in tr.get("cpu_reference", {}):
                result["gnn_epoch"]["reference_cpu_trainer_ms"] = tr["cpu_reference"]["train_step_ms_median"]
    if not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(g, x_cpu)
    if not args.no_gat:
        result["configs2_gat"] = gat_leg()
    if not args.no_sage:
        result["configs3_sage"] = sage_leg()
    if not args.no_shard_base:
        base = shard_base()
        result["weak_scaling_base"] = base
        if "roofline" in base:  # the HBM-RESIDENT number: X of the shard is 7.1 GB, far beyond L2 + Infinity Cache
            result["roofline"]["hbm_resident"] = base.pop("roofline")
    # the box's own roofs beside the spec peak: fractions of a spec number are not comparable across boxes
    try:
        from tools.papers_bench import measured_roofs

        roofs = measured_roofs(dev)
        result["roofline"].update({k: roofs[k] for k in ("measured_copy_GBs", "measured_read_GBs", "torch_copy_GBs", "kernel_copy_GBs")})
        result["roofline"]["measured_what"] = roofs["what"]
        result["roofline"]["frac_of_measured_copy"] = achieved / roofs["measured_copy_GBs"]
        hr = result["roofline"].get("hbm_resident")
        if isinstance(hr, dict) and "achieved" in hr:
            hr["frac_of_measured_copy"] = hr["achieved"] / roofs["measured_copy_GBs"]
            hr["frac_of_measured_read"] = hr["achieved"] / roofs["measured_read_GBs"]
            hr["frac_of_measured"] = hr["frac_of_measured_copy"]
    except Exception as e:
        result["roofline"]["measured_error"] = repr(e)[:300]
    if not args.no_papers:
        result["configs4_papers_1gpu"] = papers_leg(pmc=not args.no_pmc)
    return result


def papers_leg(budget_s=300, pmc=True):
    """`configs4_papers_1gpu`: BASELINE.json configs[4] at FULL size on this one GPU (tools/papers_bench.py, child
    interpreter): the papers100M-shaped graph -- 111,059,956 nodes, 1.6e9 directed / 3.2e9 symmetrised R-MAT edges generated
    on the device, F = 128 fp32 -- through csrspmm with 64-bit row pointers: forward, backward and forward + backward,
    GEdges/s, algorithmic bytes / time against the spec peak and the roofs measured on this box, peak memory.  The N = 1
    end of north_star's 1 -> 8 curve.  An error or a timeout is reported, it cannot take the line down."""
    from cogdl_amd.dist import _child_leg

    r = _child_leg([os.path.join(ROOT, "tools", "papers_bench.py"), "--steps", "3"], 6, budget_s)
    if "error" in r:
        r["error"] = r["error"][-400:]
    # HBM-side traffic of this leg's kernel, measured IN THIS RUN (north_star: "rocprof-reported achieved HBM GB/s ... in the
    # same run"): one rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass pair over a full forward pass of each graph (tools/
    # pmc_live.py collect_papers: the same seeded graph and 64-bit plan, counters summed over the pass's segment launches, ~20 s
    # per graph), calibrated on a 1 GiB copy inside the same pass.  `frac_of_measured_copy` of the traffic = counted bytes / kernel time against this
    # box's own copy roof.  The committed profile of an earlier round stays as the fallback, labelled.
    if pmc:
        from tools import pmc_live

        roofs = r.get("roofs") or {}
        for name in ("directed", "symmetrised"):
            if not isinstance(r.get(name), dict) or "error" in r[name]:
                continue
            t = pmc_live.collect_papers(name)
            if "error" not in t:
                t["frac_of_spec"] = t["hbm_GBs"] / 8000.0
                if roofs.get("measured_copy_GBs"):
                    t["frac_of_measured_copy"] = t["hbm_GBs"] / roofs["measured_copy_GBs"]
            r[name]["traffic"] = t
    measured = all(isinstance(r.get(k), dict) and "error" not in (r[k].get("traffic") or {"error": 1}) for k in ("directed", "symmetrised"))
    r["traffic_source"] = ("measured in this run (rocprofv3 --pmc over one full forward pass per graph, counters summed over its segment launches): <graph>.traffic" if measured
                           else "committed profile only (traffic_committed_profile): the in-run passes were skipped or failed")
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_papers.json")))
        r["traffic_committed_profile"] = {"source": "profiles/r05_pmc_papers.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the directed leg, "
                                                    "whole graph, round 5; NOT measured in this run -- fallback)",
                                          "hbm_side_bytes_per_pass": prof["reading"]["hbm_side_bytes_per_pass_GB"] * 1e9,
                                          "algorithmic_bytes_per_pass": prof["reading"]["algorithmic_bytes_per_pass_GB"] * 1e9}
    except Exception:
        pass
    return r


def gat_leg(budget_s=300):
    """`configs2_gat`: BASELINE.json configs[2] on this GPU -- one full-graph training step of CogDL's 2-layer GAT on the
    Reddit-shaped graph at its true size, bf16 (tools/gat_bench.py --leg, child interpreter): with the model's DEFAULT
    arguments (attn_drop 0.5, models/nn/gat.py:30) through the fused attention-dropout operator
    (install(fused_gat_dropout=True)), with attn_drop 0 through fused_gat_func, and with the default arguments on the
    unchanged layer (torch's gathers / dropout around csr_edge_softmax + csrmhspmm); `roofline` = the fractions of
    configs[2]'s kernels alone (edge_softmax forward / backward, fused GAT forward / backward with and without dropout).
    An error or a timeout is reported, it cannot take the line down."""
    from cogdl_amd.dist import _child_leg

    r = _child_leg([os.path.join(ROOT, "tools", "gat_bench.py"), "--leg", "--steps", "5"], 8, budget_s)
    if "error" in r:
        r["error"] = r["error"][-300:]
    return r


def sage_leg(budget_s=240):
    """`configs3_sage`: BASELINE.json configs[3] on this GPU -- GraphSAGE mini-batch training on the products-shaped graph,
    the whole sampled step (two sampling hops, feature gather, forward, backward, Adam) replayed as one hipGraph
    (tools/sage_bench.py --captured, 1024 seeds, child interpreter): the one-GPU base of the replica legs the N > 1 lines
    carry.  An error or a timeout is reported, it cannot take the line down."""
    from cogdl_amd.dist import _child_leg

    r = _child_leg([os.path.join(ROOT, "tools", "sage_bench.py"), "--captured", "--batch", "1024", "--steps", "50"], 7, budget_s)
    keep = ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "batch", "frontier_nodes_per_step",
            "sampled_edges_per_step", "ms_sampling_alone_rank0", "ms_feature_gather_alone_rank0", "config", "error")
    if "error" in r:
        r["error"] = r["error"][-300:]
    return {k: r[k] for k in keep if k in r}


def shard_base(budget_s=200):
    """`weak_scaling_base`: the N > 1 lines of this bench measure configs[4] (one papers100M-sized shard per GPU, weak
    scaling), a different workload from this N = 1 line (configs[1]).  So that the N > 1 values have a like-for-like
    one-GPU base, the same sharded code path is run here at world size 1 on the same shard (child interpreter, RCCL
    process group of one rank): its GEdges/s is what N x perfect scaling multiplies."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK=os.environ.get("LOCAL_RANK", "0"),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("COGDL_AMD_BASE_PORT", "29547"))
    try:
        proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--sharded", "--steps", "10", "--warmup", "2"],
                              capture_output=True, text=True, timeout=budget_s, env=env)
        line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if not line:
            return {"error": (proc.stderr or proc.stdout)[-300:]}
        r = json.loads(line[-1])
        return {"what": "vertex-sharded csr_spmm fwd+bwd at world size 1 on one papers100M-sized shard (the per-GPU work "
                        "of the N > 1 lines)", "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                "steps": r["steps"], "nodes_per_gpu": r["config"]["nodes_per_gpu"], "nnz": r["config"]["nnz_global"],
                "local_block_GEdges_s": r.get("local_block_GEdges_s_rank0"),
                "predicted": r.get("predicted"),
                "roofline": dict(r.get("roofline") or {}, what="csr_spmm over one papers100M-sized shard's local block "
                                 "(13.9 M rows, 4.1e8 edges, X = 7.1 GB): every gathered row comes from HBM")}
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % budget_s}


def bench_sharded(args):
    from cogdl_amd.dist import bench_sharded_spmm

    return bench_sharded_spmm(args)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one per GPU, torch.distributed.run
    on 127.0.0.1 -- what cogdl/trainer/trainer.py:253-274 does with mp.spawn) instead of quietly running one rank."""
    import socket
    import subprocess

    n = args.gpus
    if not args.selftest_cpu and not args.share_gpu:
        have = torch.cuda.device_count()
        if have < n:
            raise SystemExit("bench.py --gpus %d needs %d devices, this host has %d" % (n, n, have))
    # The follow-up legs and votes use MASTER_PORT + 1 .. + 8 (cogdl_amd/dist.py: _child_leg / _any_rank offsets): reserve a
    # base whose whole range is free NOW (all nine bound at once), instead of finding a collision as a leg timeout later.
    port = None
    for _ in range(64):
        with socket.socket() as probe:
            probe.bind(("127.0.0.1", 0))
            base = probe.getsockname()[1]
        if base + 8 > 65535:
            continue
        held = []
        try:
            for off in range(9):
                sk = socket.socket()
                sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                sk.bind(("127.0.0.1", base + off))
                held.append(sk)
            port = base
        except OSError:
            pass
        finally:
            for sk in held:
                sk.close()
        if port is not None:
            break
    if port is None:
        raise SystemExit("bench.py: no run of 9 free TCP ports on 127.0.0.1 for the ranks' rendezvous and the follow-up legs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--topology", default="uniform", choices=["uniform", "rmat"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-gat", action="store_true", help="skip the configs2_gat leg (GAT on the Reddit-shaped graph, bf16)")
    ap.add_argument("--no-trainer", action="store_true", help="skip the reference-Trainer epoch legs (gnn_epoch.trainer_ms)")
    ap.add_argument("--no-shard-base", action="store_true", help="skip the weak_scaling_base leg (the sharded path at world size 1)")
    ap.add_argument("--no-papers", action="store_true", help="skip the configs4_papers_1gpu leg (papers100M-shaped graph at full size on this GPU)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 --pmc passes for roofline.traffic")
    ap.add_argument("--sharded", action="store_true", help="run the N>1 workload (vertex-sharded SpMM) even at world size 1")
    ap.add_argument("--shard-nodes", type=int, default=0, help="N>1: nodes per GPU (default: papers100M/8)")
    ap.add_argument("--shard-degree", type=float, default=0.0, help="N>1: mean in-degree (default 28.8)")
    ap.add_argument("--remote-frac", type=float, default=-1.0,
                    help="N>1: fraction of a row's sources owned by other ranks (default 0.1; (N-1)/N = random partition)")
    ap.add_argument("--halo-frac", type=float, default=0.25,
                    help="N>1: halo rows per rank as a fraction of its own rows (remote sources come from boundary "
                         "regions of that total size; <= 0: uniform over the owner shard = worst-case halo)")
    ap.add_argument("--leg", default="main", choices=["main", "assumed", "worst"], help="(internal) which leg a child interpreter runs")
    ap.add_argument("--papers-scale", type=int, default=0,
                    help="N>1 main leg: shard the papers100M-shaped graph at 1/this scale (default: 1 = full size; the "
                         "self-test modes default to small graphs)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N>1: only the main line (no worst-case partition, no configs[3] leg)")
    ap.add_argument("--legs-budget-s", type=float, default=420.0,
                    help="N>1: no follow-up leg is started once this many seconds have passed (all ranks decide together)")
    ap.add_argument("--no-sage", action="store_true", help="skip the configs[3] GraphSAGE leg (N=1: configs3_sage, N>1: the replica leg)")
    ap.add_argument("--worst-case-scale", type=int, default=4, help="N>1: the worst-case leg's shards are 1/this the size")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher self-test: gloo ranks on the host with libcogdl_host kernels and tiny shards (tests only; "
                         "its numbers mean nothing)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="orchestration smoke test on a one-GPU box: all ranks on cuda:0, rows exchanged through gloo (its "
                         "numbers are not multi-GPU numbers)")
    args = ap.parse_args()
    args.t0 = T0
    args.bench_script = os.path.abspath(__file__)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if not args.selftest_cpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.sharded and args.gpus == 1):
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: refusing to report a %d-GPU line from %d rank(s)"
                         % (args.gpus, world, args.gpus, world))
    if args.gpus > 1 or world > 1 or args.sharded:
        result = bench_sharded(args)
    else:
        result = bench_single(args)
    if result is not None:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
