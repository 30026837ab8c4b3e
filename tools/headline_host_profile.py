#!/usr/bin/env python3
"""Where the headline step's time goes (bench.py: bench_single.step = the reference dispatcher's call, fresh `.int()` copies of
the structure, csrspmm forward + backward through autograd on the arxiv-sized graph, F = 128 fp32).

    python tools/headline_host_profile.py              host enqueue time vs wall time per step, cProfile by own time
    python tools/headline_host_profile.py --trace      no cProfile: 10 + 100 steps for `rocprofv3 --kernel-trace`
    python tools/headline_host_profile.py --analyse <kernel_trace.csv>   kernels per step, busy time, gaps

The two SpMM launches are 2 x 0.178 ms; a step the host enqueues in less than that is bound by the GPU."""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(path, steps=100):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    main = [i for i, r in enumerate(rows) if "rowreduce_main_kernel" in r[2]]
    per = len(main) // (steps + 10)
    assert per >= 2, "expected two SpMM launches per step"
    first = main[len(main) - per * steps]
    # a step = from its first SpMM-preceding kernel; simpler: cut at every `per`-th main kernel and attribute what lies between
    sel = rows[first:]
    span = (sel[-1][1] - sel[0][0]) / 1e3 / steps
    busy, by = 0.0, {}
    for s, e, n in sel:
        busy += (e - s) / 1e3
        k = n.split("<")[0].split("(")[0][:70]
        by[k] = by.get(k, 0.0) + (e - s) / 1e3
    print("per step (last %d steps): span %.1f us, kernels busy %.1f us, gaps %.1f us, %.1f kernels" % (steps, span, busy / steps, span - busy / steps, len(sel) / steps))
    for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print("  %-72s %8.1f us / step" % (k, v / steps))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        return analyse(sys.argv[2])
    import torch

    from cogdl_amd import synth
    from cogdl_amd.operators.spmm import csrspmm

    dev = torch.device("cuda:0")
    g = synth.arxiv_like(seed=0)
    gd = g.to(dev)
    r64, c64 = gd.rowptr.long(), gd.colind.long()
    x = torch.randn(g.num_nodes, 128, device=dev).requires_grad_()
    gout = torch.randn(g.num_nodes, 128, device=dev)

    def step():
        out = csrspmm(r64.int(), c64.int(), x, gd.weight, True)
        x.grad = None
        out.backward(gout)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    if "--trace" in sys.argv:
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        return
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("host enqueue %.1f us/step, wall %.1f us/step" % (t_host / 100 * 1e6, t_all / 100 * 1e6))
    # the host alone: the same calls on a graph so small that the GPU is never the bottleneck
    n_s = 2048
    rs = (torch.arange(n_s + 1, device=dev) * 8).long()
    cs = torch.randint(0, n_s, (n_s * 8,), device=dev).long()
    ws = torch.rand(n_s * 8, device=dev)
    xs = torch.randn(n_s, 128, device=dev).requires_grad_()
    gos = torch.randn_like(xs)

    def small():
        out = csrspmm(rs.int(), cs.int(), xs, ws, True)
        xs.grad = None
        out.backward(gos)

    for _ in range(10):
        small()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        small()
    torch.cuda.synchronize()
    print("2048-row graph: wall %.1f us/step (the host's own pace)" % ((time.perf_counter() - t0) / 200 * 1e6))
    from cogdl_amd.operators import spmm as spmm_mod

    for name, make in (("a plain list (events created per launch)", lambda: []), ("KernelEventLog(every = 8 launches)", lambda: spmm_mod.KernelEventLog(every=8))):
        spmm_mod.KERNEL_EVENTS = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            small()
        torch.cuda.synchronize()
        print("2048-row graph with KERNEL_EVENTS = %s: wall %.1f us/step" % (name, (time.perf_counter() - t0) / 200 * 1e6))
        spmm_mod.KERNEL_EVENTS = None
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
