#!/usr/bin/env python3
"""Where one epoch of bench.py's default GCN goes (bench.py `gnn_epoch.accounting`; round-5 verdict, item 5): one
`rocprofv3 --kernel-trace` of the hipGraph replay of the training step, folded into

    library_kernel_ms   kernels of libcogdl_hip (csr_spmm x 4, MFMA linear forward / grad_input / weight gradient, ...)
    torch_kernel_ms     everything else on the stream (cross_entropy's log_softmax / nll_loss, dropout, relu, Adam, copies),
                        with the three largest by name
    host_gap_ms         the replay window minus the time a kernel was running: launch gaps inside the hipGraph

per step, over REPLAYS replays that follow a pause in the timeline (so that they can be told from capture and warm-up).
The metric's second half ("epoch time = wall time of one Trainer.train_step") is 1-2 ms of which the
library's kernels are a fraction: the line should show that, not leave it to prose.

    python tools/epoch_account.py probe      the workload (run under rocprofv3 by collect())
    python tools/epoch_account.py collect    prints the JSON collect() returns

Measurement infrastructure, not product code."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAYS = 20
PAUSE_S = 0.25


def probe():
    import torch

    sys.path.insert(0, ROOT)
    import bench
    from cogdl_amd import synth

    dev = torch.device("cuda:0")
    g = synth.arxiv_like(seed=0, topology="uniform")
    gd = g.to(dev)
    x = torch.randn(g.num_nodes, 128, generator=torch.Generator().manual_seed(0)).to(dev)
    step = bench.gcn_epoch_ms(gd, gd.rowptr.long(), gd.colind.long(), x, reps=3, warmup=2, mfma_linear=True, captured=True,
                              return_step=True)
    torch.cuda.synchronize()
    time.sleep(PAUSE_S)
    t0 = time.perf_counter()
    for _ in range(REPLAYS):
        step()
    torch.cuda.synchronize()
    print("PROBE " + json.dumps({"replays": REPLAYS, "wall_ms_per_step_in_probe": (time.perf_counter() - t0) / REPLAYS * 1e3}))


def _short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    for cut in ("<", "("):
        if cut in name:
            name = name[:name.index(cut)]
    return name[-80:]


def fold(trace_csv, replays=REPLAYS):
    rows = []
    for r in csv.DictReader(open(trace_csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if not rows:
        return {"error": "empty kernel trace"}
    # the timed replays: everything after the LAST pause of more than PAUSE_S / 2 between two kernels
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][0] - rows[i - 1][1] > PAUSE_S * 0.5e9:
            cut = i
    rows = rows[cut:]
    if len(rows) % replays != 0:
        return {"error": "%d kernels after the pause do not divide into %d replays" % (len(rows), replays)}
    lib = torch_ns = 0
    by_name = {}
    for s, e, name in rows:
        d = e - s
        if "cogdl::" in name:
            lib += d
        else:
            torch_ns += d
            k = _short(name)
            by_name[k] = by_name.get(k, 0) + d
    window = rows[-1][1] - rows[0][0]
    top = sorted(by_name.items(), key=lambda t: -t[1])[:3]
    return {"kernels_per_step": len(rows) // replays, "replays": replays,
            "library_kernel_ms": lib / replays / 1e6, "torch_kernel_ms": torch_ns / replays / 1e6,
            "host_gap_ms": (window - lib - torch_ns) / replays / 1e6, "step_ms_in_trace": window / replays / 1e6,
            "torch_top3": [{"kernel": k, "ms": v / replays / 1e6} for k, v in top]}


def collect(timeout_s=150):
    """-> {"library_kernel_ms", "torch_kernel_ms", "torch_top3", "host_gap_ms", "step_ms_in_trace", ...} or {"error": ...};
    never raises."""
    try:
        exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        if not os.path.exists(exe):
            return {"error": "rocprofv3 not found"}
        tmp = tempfile.mkdtemp(prefix="cogdl_epoch_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            proc = subprocess.run([exe, "--kernel-trace", "-f", "csv", "-d", tmp, "-o", "ep", "--", sys.executable,
                                   os.path.abspath(__file__), "probe"], cwd="/tmp", env=env, capture_output=True, text=True,
                                  timeout=timeout_s)
            files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
            if not files:
                return {"error": "no kernel trace (rc %d): %s" % (proc.returncode, (proc.stderr or proc.stdout)[-300:])}
            out = fold(files[0])
            for ln in (proc.stdout or "").splitlines():
                if ln.startswith("PROBE "):
                    out["wall_ms_per_step_in_probe"] = json.loads(ln[6:])["wall_ms_per_step_in_probe"]
            out["source"] = ("rocprofv3 --kernel-trace over tools/epoch_account.py probe (the hipGraph replay of bench.py's default GCN "
                             "step, MFMA linear on), %d replays after a pause, in the same run as this line" % REPLAYS)
            return out
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    except Exception as e:
        return {"error": "epoch_account.collect: %r" % (e,)}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "probe":
        probe()
    elif len(sys.argv) > 2 and sys.argv[1] == "fold":
        print(json.dumps(fold(sys.argv[2]), indent=1))
    else:
        print(json.dumps(collect(), indent=1))
