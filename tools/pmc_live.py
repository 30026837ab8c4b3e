#!/usr/bin/env python3
"""HBM-side traffic of the bench's csr_spmm kernel, measured in the SAME run as the bench line (bench.py's
`roofline.traffic`): `collect()` runs `rocprofv3 --kernel-trace --pmc <counter>` over this file's own probe workload,
one pass per counter (FETCH_SIZE, WRITE_SIZE -- never combined with other trace domains), and folds the CSVs into
bytes per launch, calibrated on a 1 GiB device copy inside the same pass exactly as MI355X_MICROARCH.md's HBM section
prescribes for gfx950 (FETCH_SIZE under-reports wide reads by 2x there; the factor used is the one measured).

    python tools/pmc_live.py probe <uniform|rmat> <feat>     the workload (run under rocprofv3 by collect())
    python tools/pmc_live.py collect [uniform|rmat] [feat]   prints the JSON collect() returns

Measurement infrastructure, not product code.  What the counters see: L2 -> fabric requests, so Infinity-Cache (MALL)
hits are included -- for the arxiv-sized graph (X = 87 MB) this is fabric traffic, not DRAM bursts."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GIB = float(1 << 30)
LAUNCHES = 10


def probe(topology, feat):
    import torch

    sys.path.insert(0, ROOT)
    from cogdl_amd import synth
    from cogdl_amd.operators.spmm import csr_spmm_raw

    dev = "cuda:0"
    a = torch.randn(256 * 1024 * 1024, device=dev)  # calibration: 1 GiB read + 1 GiB written, far beyond the 256 MiB MALL
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    del a, b
    g = synth.arxiv_like(seed=0, topology=topology).to(dev)
    x = torch.randn(g.num_nodes, feat, device=dev)
    for _ in range(LAUNCHES):
        csr_spmm_raw(g.rowptr, g.colind, g.weight, x)
    torch.cuda.synchronize()
    print("probe done", g.nnz)


def _fold(path):
    by = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        d = by.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                    "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
        d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    return [by[k] for k in sorted(by)]


def collect(topology="uniform", feat=128, timeout_s=100):
    """-> {"hbm_bytes_per_launch": ..., "read": ..., "write": ..., "kernel_us_profiled": ..., "calibration": {...}}
    or {"error": "..."}; never raises (a rocprofv3 with another CSV schema, a truncated file, ... must not take the bench
    line down: it then falls back to the committed profile and says so)."""
    try:
        return _collect(topology, feat, timeout_s)
    except Exception as e:
        return {"error": "pmc_live.collect: %r" % (e,)}


def _collect(topology, feat, timeout_s):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="cogdl_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over tools/pmc_live.py "
                     "probe, in the same run as this line; units calibrated on a 1 GiB copy in the same pass",
           "topology": topology, "feat": feat}
    per, cal, durs = {}, {}, []
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            try:
                proc = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", d, "-o", "pmc", "--",
                                       sys.executable, os.path.abspath(__file__), "probe", topology, str(feat)],
                                      cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {"error": "rocprofv3 --pmc %s timed out after %d s" % (counter, timeout_s)}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {"error": "rocprofv3 --pmc %s produced no counter CSV (rc %d): %s"
                                 % (counter, proc.returncode, (proc.stderr or proc.stdout)[-300:])}
            copies, mains = [], []
            for f in files:
                for disp in _fold(f):
                    v = disp["c"].get(counter)
                    if v is None:
                        continue
                    if "copyBuffer" in disp["name"]:
                        copies.append(v)
                    elif "rowreduce_main_kernel" in disp["name"] and "SpmmOp" in disp["name"]:
                        mains.append(v)
                        durs.append((disp["t"][1] - disp["t"][0]) / 1e3)
            copies = [v for v in copies if v >= 0.5 * max(copies)] if copies else copies  # the 1 GiB copies only
            if not copies or not mains:
                return {"error": "no %s samples for the calibration copy / the csr_spmm kernel" % counter}
            cal[counter] = GIB / (sum(copies) / len(copies))
            per[counter] = sum(mains) / len(mains) * cal[counter]
            out["launches_profiled"] = len(mains)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out["calibration"] = {"fetch_bytes_per_unit": cal["FETCH_SIZE"], "write_bytes_per_unit": cal["WRITE_SIZE"]}
    out["read"], out["write"] = per["FETCH_SIZE"], per["WRITE_SIZE"]
    out["hbm_bytes_per_launch"] = per["FETCH_SIZE"] + per["WRITE_SIZE"]
    out["kernel_us_profiled"] = sum(durs) / len(durs)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "probe":
        probe(sys.argv[2] if len(sys.argv) > 2 else "uniform", int(sys.argv[3]) if len(sys.argv) > 3 else 128)
    else:
        print(json.dumps(collect(sys.argv[2] if len(sys.argv) > 2 else "uniform",
                                 int(sys.argv[3]) if len(sys.argv) > 3 else 128), indent=1))
