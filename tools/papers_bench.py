#!/usr/bin/env python3
"""bench.py's `configs4_papers_1gpu` leg: BASELINE.json configs[4] (GCN aggregation on ogbn-papers100M: 111 M nodes,
1.6e9 directed edges, 128-wide fp32 features) at FULL size on ONE MI355X -- the N = 1 end of north_star's 1 -> 8 curve.

Two graphs from the same R-MAT pairs (cogdl_amd/synth.py: papers100m_like), both through `csrspmm` with 64-bit row
pointers (cogdl_amd/bigcsr.py: row segments of ~2^29 edges on the 32-bit kernels):
  directed      1,615,685,872 edges (the dataset's citation pairs; row = aggregation target)
  symmetrised   3,231,371,744 edges (> 2^31) -- what CogDL feeds GCN (cogdl/datasets/ogb.py:50-55)
each: csr_spmm forward alone (HIP-event timed over `steps` launches) and forward + backward (grad_out := out, so three
N x 128 fp32 tensors = 171 GB are live, not four); GEdges/s, algorithmic bytes / time against the 8 TB/s spec and the
copy roof measured on this box, peak allocated memory.  One JSON line on stdout."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def b_alg(nnz, m, f):
    return nnz * (4 + 4 + f * 4) + m * (4 + f * 4)


def measured_roofs(dev, gib=1):
    """The box's own roofs: a 1 GiB device-to-device copy (read + write bytes; torch's copy_ and this
    library's 16-byte-vector copy kernel, the better of the two) and a read-only 16-byte stream over the same buffer,
    best of 7, far beyond L2 + Infinity Cache."""
    from cogdl_amd import _lib

    n = gib * (1 << 30) // 4
    a = torch.randn(n, device=dev)
    b = torch.empty_like(a)
    sink = torch.empty(4096 * 4, dtype=torch.int32, device=dev)
    lib = _lib.hip()
    best = {"torch_copy": 1e9, "copy": 1e9, "read": 1e9}
    for _ in range(7):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        b.copy_(a)
        ev[1].record()
        rc1 = lib.cogdl_hip_probe_copy_stream(a.data_ptr(), b.data_ptr(), n * 4, _lib.stream_of(a))
        ev[2].record()
        rc2 = lib.cogdl_hip_probe_read_stream(a.data_ptr(), n * 4, sink.data_ptr(), _lib.stream_of(a))
        ev[3].record()
        torch.cuda.synchronize()
        _lib.check(rc1, "probe_copy_stream")
        _lib.check(rc2, "probe_read_stream")
        for key, i in (("torch_copy", 0), ("copy", 1), ("read", 2)):
            best[key] = min(best[key], ev[i].elapsed_time(ev[i + 1]))
    gbs = {k: (2 if "copy" in k else 1) * n * 4 / (v * 1e-3) / 1e9 for k, v in best.items()}
    return {"measured_copy_GBs": max(gbs["copy"], gbs["torch_copy"]), "measured_read_GBs": gbs["read"],
            "torch_copy_GBs": gbs["torch_copy"], "kernel_copy_GBs": gbs["copy"],
            "what": "%d GiB device-to-device copy (read + written bytes; torch copy_ / cogdl_hip_probe_copy_stream, the better) and a "
                    "read-only 16-byte-vector stream (cogdl_hip_probe_read_stream), best of 7" % gib}


def run(g, feat, steps, dev, roofs):
    from cogdl_amd.bigcsr import clear_plans, plan_of
    from cogdl_amd.operators.spmm import csrspmm

    n = g.num_nodes
    torch.cuda.reset_peak_memory_stats()
    x = torch.randn(n, feat, device=dev)
    t0 = time.perf_counter()
    plan = plan_of(g.rowptr, g.colind, n)
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    res = {"nodes": n, "nnz": g.nnz, "feat": feat, "segments": plan.n_segments, "plan_s": plan_s}
    with torch.no_grad():
        out = plan.spmm(g.weight, x)  # warm-up
        del out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = plan.spmm(g.weight, x)
            del out
        e1.record()
        torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / steps
    bytes_alg = b_alg(g.nnz, n, feat)
    ach = bytes_alg / (fwd_ms * 1e-3) / 1e9
    res["forward"] = {"ms": fwd_ms, "GEdges_s": g.nnz / (fwd_ms * 1e-3) / 1e9, "algorithmic_bytes": bytes_alg, "achieved_GBs": ach,
                      "frac": ach / HBM_PEAK_GBS, "frac_of_measured_read": ach / roofs["measured_read_GBs"],
                      "frac_of_measured_copy": ach / roofs["measured_copy_GBs"]}
    # forward + backward through the autograd operator (the transpose is plan time: built by the first backward, cached)
    x.requires_grad_()
    t0 = time.perf_counter()
    out = csrspmm(g.rowptr, g.colind, x, g.weight, True)
    out.backward(out.detach())
    torch.cuda.synchronize()
    res["first_step_with_transpose_s"] = time.perf_counter() - t0
    del out
    x.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = csrspmm(g.rowptr, g.colind, x, g.weight, True)
        out.backward(out.detach())
        del out
        x.grad = None
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    ach2 = 2 * bytes_alg / (step_ms * 1e-3) / 1e9
    res["forward_backward"] = {"ms": step_ms, "GEdges_s": 2 * g.nnz / (step_ms * 1e-3) / 1e9, "achieved_GBs": ach2,
                               "frac": ach2 / HBM_PEAK_GBS, "frac_of_measured_read": ach2 / roofs["measured_read_GBs"]}
    # the backward SpMM alone (A^T on the cached 64-bit transpose), event timed like the forward
    t, w_t = plan.transposed(g.weight)
    xd = x.detach()
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            out = t.spmm(w_t, xd)
            del out
        e1.record()
        torch.cuda.synchronize()
    bwd_ms = e0.elapsed_time(e1) / steps
    res["backward_alone"] = {"ms": bwd_ms, "achieved_GBs": bytes_alg / (bwd_ms * 1e-3) / 1e9,
                             "frac": bytes_alg / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "segments": t.n_segments}
    res["peak_allocated_GB"] = torch.cuda.max_memory_allocated() / 1e9
    del t, w_t, xd
    del x, plan
    clear_plans()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--pairs", type=int, default=0)
    ap.add_argument("--only", default="", choices=["", "directed", "symmetrised"])
    args = ap.parse_args()
    from cogdl_amd import synth

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    nodes, pairs = args.nodes or synth.PAPERS_NODES, args.pairs or synth.PAPERS_PAIRS
    roofs = measured_roofs(dev)
    t0 = time.perf_counter()
    src, dst = synth.rmat_pairs_i32(nodes, pairs, 0, dev)
    torch.cuda.synchronize()
    result = {"what": "csr_spmm on the papers100M-shaped graph at full size on ONE GPU (BASELINE configs[4], the N = 1 end of the "
                      "scaling curve), fp32, R-MAT pairs generated on the device, multi-edges KEPT (the reference coalesces them, "
                      "cogdl/datasets/ogb.py:50-55 -- R-MAT draws duplicates, the dataset has none: the edge count is the pair count; harmless for "
                      "bandwidth), sym-normalised weights",
              "roofs": roofs, "pairs_s": time.perf_counter() - t0, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    for name, sym in (("directed", False), ("symmetrised", True)):
        if args.only and args.only != name:
            continue
        try:
            t0 = time.perf_counter()
            g = synth.big_csr_from_pairs(src, dst, nodes, sym)
            torch.cuda.synchronize()
            build_s = time.perf_counter() - t0
            if sym:
                del src, dst
                src = dst = None
            torch.cuda.empty_cache()
            r = run(g, args.feat, args.steps, dev, roofs)
            r["csr_build_s"] = build_s
            result[name] = r
            del g
            torch.cuda.empty_cache()
        except Exception as e:  # one graph failing (e.g. out of memory on a smaller device) must not lose the other
            result[name] = {"error": repr(e)[:400]}
            torch.cuda.empty_cache()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
