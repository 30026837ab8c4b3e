#!/usr/bin/env python3
"""Fold rocprofv3 --pmc CSVs (one directory per counter pass under <dir>/<prefix>*/) into per-kernel means:
calibrated HBM-side bytes (FETCH_SIZE / WRITE_SIZE scaled on the 1 GiB copy the probe starts with, as
MI355X_MICROARCH.md prescribes for gfx950), launches and mean duration.  Kernels are keyed by their (shortened) name in
dispatch-order groups, so that e.g. the forward and backward instantiations stay apart.
usage: pmc_by_kernel.py <dir> <prefix> [name-substring ...]"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, prefix = sys.argv[1], sys.argv[2]
want = sys.argv[3:] or ["es_flat", "SpmmOp"]
GIB = float(1 << 30)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
calib = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, prefix + "*", "**", "*counter_collection.csv"), recursive=True)):
    by_dispatch = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        d = by_dispatch.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                             "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
        d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    for did in sorted(by_dispatch):
        d = by_dispatch[did]
        name = d["name"]
        if "copyBuffer" in name:
            for c, v in d["c"].items():
                calib[c].append(v)
            continue
        if not any(w in name for w in want):
            continue
        short = re.sub(r"\(.*", "", name).replace("void cogdl::", "").replace("cogdl::", "").replace("esf::", "")
        for c, v in d["c"].items():
            vals[short][c].append(v)
        durs[short].append((d["t"][1] - d["t"][0]) / 1e3)
mean = lambda l: sum(l) / len(l) if l else None  # noqa: E731
out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one pass per counter), workload tools/pmc_probe_es.py",
       "calibration": {}, "kernels": {}}
def big(lst):  # the 1 GiB calibration copies: the FIRST three sizeable copyBuffer dispatches of a probe (later ones may be
    # larger -- a 3.3 GB torch.cat of a shard generator skewed the calibration of round 3's first shard profile)
    first = [v for v in lst if v >= 0.05 * max(lst)][:3] if lst else lst
    return first


f, w = mean(big(calib["FETCH_SIZE"])), mean(big(calib["WRITE_SIZE"]))
if f:
    out["calibration"]["fetch_bytes_per_unit"] = GIB / f
if w:
    out["calibration"]["write_bytes_per_unit"] = GIB / w
out["calibration"]["note"] = "units calibrated on the probe's 1 GiB copy (1 GiB read + 1 GiB written)"
for k, cs in vals.items():
    e = {c: mean(v) for c, v in cs.items()}
    e["launches"] = max(len(v) for v in cs.values())
    e["duration_us_profiled"] = mean(durs[k])
    if "FETCH_SIZE" in e and f:
        e["hbm_read_bytes"] = e["FETCH_SIZE"] * GIB / f
    if "WRITE_SIZE" in e and w:
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * GIB / w
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if e.get("duration_us_profiled"):
            e["hbm_GBs"] = e["hbm_bytes_per_launch"] / e["duration_us_profiled"] / 1e3
            e["hbm_frac_of_8TBs"] = e["hbm_GBs"] / 8000.0
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
