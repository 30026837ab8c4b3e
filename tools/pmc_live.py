#!/usr/bin/env python3
"""HBM-side traffic of the bench's csr_spmm kernel, measured in the SAME run as the bench line (bench.py's
`roofline.traffic`): `collect()` runs `rocprofv3 --kernel-trace --pmc <counter>` over this file's own probe workload,
one pass per counter (FETCH_SIZE, WRITE_SIZE -- never combined with other trace domains), and folds the CSVs into
bytes per launch, calibrated on a 1 GiB device copy inside the same pass exactly as MI355X_MICROARCH.md's HBM section
prescribes for gfx950 (FETCH_SIZE under-reports wide reads by 2x there; the factor used is the one measured).

    python tools/pmc_live.py probe <uniform|rmat> <feat>     the workload (run under rocprofv3 by collect())
    python tools/pmc_live.py collect [uniform|rmat] [feat]   prints the JSON collect() returns
    python tools/pmc_live.py collect_papers <directed|symmetrised>   the same for one row segment of the papers100M-shaped graph

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


PAPERS_PASSES = 2


def probe_papers(which, feat):
    """The papers100M-shaped graph of bench.py's `configs4_papers_1gpu` leg at FULL size under the counters: the same seeded
    graph (synth.papers100m_like), the same 64-bit plan (cogdl_amd/bigcsr.py: row segments of ~2^29 edges on the 32-bit
    kernels), PAPERS_PASSES forward passes -- the counters of a pass are the sum over its segment launches.  The operand is
    57 GB: far beyond every cache."""
    import torch

    sys.path.insert(0, ROOT)
    from cogdl_amd import synth
    from cogdl_amd.bigcsr import plan_of

    dev = "cuda:0"
    a = torch.randn(256 * 1024 * 1024, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    del a, b
    g = synth.papers100m_like(dev, symmetrise=which == "symmetrised", seed=0)
    x = torch.randn(g.num_nodes, feat, device=dev)
    plan = plan_of(g.rowptr, g.colind, g.num_nodes)
    with torch.no_grad():
        out = plan.spmm(g.weight, x)  # warm-up (not counted: see _collect's `skip`)
        del out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(PAPERS_PASSES):
            out = plan.spmm(g.weight, x)
            del out
        e1.record()
        torch.cuda.synchronize()
    print("PROBE " + json.dumps({"which": which, "segments": plan.n_segments, "rows": g.num_nodes, "nnz": g.nnz, "feat": feat,
                                 "passes": PAPERS_PASSES, "pass_ms_event_timed": e0.elapsed_time(e1) / PAPERS_PASSES}))


def _fold(path):
    by = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        d = by.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                    "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
        d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    return [by[k] for k in sorted(by)]


def collect_papers(which="directed", feat=128, timeout_s=240):
    """HBM-side traffic of ONE forward pass over the papers100M-shaped graph at full size (`which`: directed | symmetrised),
    measured in the same run as the bench line: the counters summed over the pass's segment launches, the pass's duration
    under the profiler (HIP events around the passes: counter collection serialises the launches but measured no slower), its
    algorithmic bytes and the rate `hbm_GBs` = counted bytes / that duration.  Never raises."""
    try:
        out = _collect(None, feat, timeout_s, ["probe_papers", which, str(feat)], per_pass=True)
        if "error" in out:
            return out
        seg = out.get("probe") or {}
        if not seg:
            return {"error": "the papers probe printed no PROBE line"}
        b_alg = seg["nnz"] * (4 + 4 + feat * 4) + seg["rows"] * (4 + feat * 4)
        out["algorithmic_bytes_per_pass"] = b_alg
        out["hbm_bytes_per_pass"] = out.pop("hbm_bytes_per_launch")
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_pass"] / b_alg
        out["pass_ms_under_profiler"] = seg.pop("pass_ms_event_timed")
        out["kernel_ms_per_pass_profiled"] = out.pop("kernel_us_profiled") / 1e3
        out["hbm_GBs"] = out["hbm_bytes_per_pass"] / (out["kernel_ms_per_pass_profiled"] * 1e-3) / 1e9
        out["hbm_GBs_what"] = "counted bytes per pass / the summed durations of the pass's csr_spmm kernels in the same profiled run"
        out["what"] = ("one forward pass (%s segment launches) over the %s papers100M-shaped graph (%s rows, %s edges) under "
                       "rocprofv3 --pmc, in the same run as this line" % (seg.get("segments"), which, seg.get("rows"), seg.get("nnz")))
        return out
    except Exception as e:
        return {"error": "pmc_live.collect_papers: %r" % (e,)}


def collect(topology="uniform", feat=128, timeout_s=100):
    """-> {"hbm_bytes_per_launch": ..., "read": ..., "write": ..., "kernel_us_profiled": ..., "calibration": {...}}
    or {"error": "..."}; never raises (a rocprofv3 with another CSV schema, a truncated file, ... must not take the bench
    line down: it then falls back to the committed profile and says so)."""
    try:
        return _collect(topology, feat, timeout_s)
    except Exception as e:
        return {"error": "pmc_live.collect: %r" % (e,)}


def _collect(topology, feat, timeout_s, probe_argv=None, per_pass=False):
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
                                       sys.executable, os.path.abspath(__file__)] + (probe_argv or ["probe", topology, str(feat)]),
                                      cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {"error": "rocprofv3 --pmc %s timed out after %d s" % (counter, timeout_s)}
            for ln in (proc.stdout or "").splitlines():
                if ln.startswith("PROBE "):
                    out["probe"] = json.loads(ln[6:])
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {"error": "rocprofv3 --pmc %s produced no counter CSV (rc %d): %s"
                                 % (counter, proc.returncode, (proc.stderr or proc.stdout)[-300:])}
            copies, mains, d_this = [], [], []
            for f in files:
                for disp in _fold(f):
                    v = disp["c"].get(counter)
                    if v is None:
                        continue
                    if "copyBuffer" in disp["name"]:
                        copies.append(v)
                    elif "rowreduce_main_kernel" in disp["name"] and "SpmmOp" in disp["name"]:
                        mains.append(v)
                        d_this.append((disp["t"][1] - disp["t"][0]) / 1e3)
            # the 1 GiB calibration copies only: the FIRST three sizeable copyBuffer dispatches (a probe that builds a big graph
            # may copy larger buffers later -- they would skew the unit)
            copies = [v for v in copies if v >= 0.05 * max(copies)][:3] if copies else copies
            if not copies or not mains:
                return {"error": "no %s samples for the calibration copy / the csr_spmm kernel" % counter}
            cal[counter] = GIB / (sum(copies) / len(copies))
            if per_pass:  # a pass = n_seg launches; the first pass of the probe is its warm-up
                n_pass = (out.get("probe") or {}).get("passes")
                n_seg = (out.get("probe") or {}).get("segments")
                if not n_pass or not n_seg or len(mains) != (n_pass + 1) * n_seg:
                    return {"error": "papers probe: %d csr_spmm launches for %s passes of %s segments (+ warm-up)" % (len(mains), n_pass, n_seg)}
                per[counter] = sum(mains[n_seg:]) / n_pass * cal[counter]
                durs.append(sum(d_this[n_seg:]) / n_pass)
            else:
                per[counter] = sum(mains) / len(mains) * cal[counter]
                durs.extend(d_this)
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
    elif len(sys.argv) > 1 and sys.argv[1] == "probe_papers":
        probe_papers(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 128)
    elif len(sys.argv) > 1 and sys.argv[1] == "collect_papers":
        print(json.dumps(collect_papers(sys.argv[2] if len(sys.argv) > 2 else "directed"), indent=1))
    else:
        print(json.dumps(collect(sys.argv[2] if len(sys.argv) > 2 else "uniform",
                                 int(sys.argv[3]) if len(sys.argv) > 3 else 128), indent=1))
