#!/usr/bin/env python3
"""Kernel names: rowreduce_main_kernel<SpmmOp<...>> = the csr_spmm launch (row blocks + long-row workgroups),
rowreduce_combine_kernel = the per-long-row merge.

Fold the rocprofv3 --pmc CSVs under <dir>/pmc_*/ (one directory per counter pass, all running
tools/pmc_probe.py) into one JSON: calibration of FETCH_SIZE / WRITE_SIZE on a 1 GiB copy, then per workload
phase of the probe the mean counters per csr_spmm launch, the calibrated HBM-side bytes and the L2 hit rate.

The probe launches csr_spmm in a fixed schedule (PHASES below); dispatches of the main kernel are attributed to a
phase by their ordinal among the main kernel's dispatches (rocprofv3 keeps dispatch order)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
GIB = float(1 << 30)
# (phase name, launches) -- must match tools/pmc_probe.py
PHASES = [("arxiv_uniform_F128", 10), ("arxiv_rmat_F128", 10), ("scaled_4M_nodes_F128", 5)]
KERNELS = {"rowreduce_main_kernel": "main", "rowreduce_combine_kernel": "combine"}


def phase_of(ordinal):
    for name, n in PHASES:
        if ordinal < n:
            return name
        ordinal -= n
    return None


def kernel_of(name):
    for key, short in KERNELS.items():
        if key in name:
            return short
    return None


calib = collections.defaultdict(list)                                  # (kernel, counter) -> values
per = collections.defaultdict(lambda: collections.defaultdict(list))   # phase -> (kernel, counter) -> values
durs = collections.defaultdict(lambda: collections.defaultdict(list))  # phase -> kernel -> us
for path in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    by_dispatch = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        d = by_dispatch.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                             "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
        d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    ordinal = collections.defaultdict(int)
    for did in sorted(by_dispatch):
        d = by_dispatch[did]
        name = d["name"]
        if "copyBuffer" in name or "reduce_kernel" in name:
            tag = "copy_1GiB" if "copyBuffer" in name else "sum_1GiB"
            for c, v in d["c"].items():
                calib[(tag, c)].append(v)
            continue
        k = kernel_of(name)
        if k is None:
            continue
        ph = phase_of(ordinal[k])
        ordinal[k] += 1
        if ph is None:
            continue
        for c, v in d["c"].items():
            per[ph][(k, c)].append(v)
        durs[ph][k].append((d["t"][1] - d["t"][0]) / 1e3)


def mean(lst):
    return sum(lst) / len(lst) if lst else None


out = {"source": "rocprofv3 --kernel-trace --pmc <counter>, one pass per counter group, workload tools/pmc_probe.py",
       "calibration": {}, "phases": {}}
cal = out["calibration"]
f_copy, w_copy, f_sum = mean(calib[("copy_1GiB", "FETCH_SIZE")]), mean(calib[("copy_1GiB", "WRITE_SIZE")]), mean(calib[("sum_1GiB", "FETCH_SIZE")])
if f_copy:
    cal["FETCH_SIZE_units_for_1GiB_read(copy)"] = f_copy
    cal["fetch_bytes_per_unit"] = GIB / f_copy
if f_sum:
    cal["FETCH_SIZE_units_for_1GiB_read(sum)"] = f_sum
    cal.setdefault("fetch_bytes_per_unit", GIB / f_sum)
if w_copy:
    cal["WRITE_SIZE_units_for_1GiB_written"] = w_copy
    cal["write_bytes_per_unit"] = GIB / w_copy
for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"):
    v = mean(calib[("copy_1GiB", c)])
    if v:
        cal["bytes_per_%s(copy)" % c] = GIB / v
cal["note"] = ("FETCH_SIZE/WRITE_SIZE are nominally KiB.  MI355X_MICROARCH.md (HBM section): gfx950 FETCH_SIZE reports half "
               "of a wide coalesced read, i.e. ~2048 bytes per unit; the factor used below is the one measured here on "
               "the 1 GiB copy.  Infinity-Cache hits are counted (these are L2 -> fabric requests, not DRAM bursts).")
for ph, kc in per.items():
    e = {}
    for (k, c), lst in kc.items():
        e.setdefault(k, {})[c] = mean(lst)
    for k, lst in durs[ph].items():
        e.setdefault(k, {})["duration_us_profiled"] = mean(lst)
        e[k]["launches"] = len(lst)
    m = e.get("main", {})
    if "FETCH_SIZE" in m and "fetch_bytes_per_unit" in cal:
        m["hbm_read_bytes"] = m["FETCH_SIZE"] * cal["fetch_bytes_per_unit"]
    if "WRITE_SIZE" in m and "write_bytes_per_unit" in cal:
        m["hbm_write_bytes"] = m["WRITE_SIZE"] * cal["write_bytes_per_unit"]
    if "hbm_read_bytes" in m and "hbm_write_bytes" in m:
        m["hbm_bytes_per_launch"] = m["hbm_read_bytes"] + m["hbm_write_bytes"]
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        m["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out["phases"][ph] = e
print(json.dumps(out, indent=1))
