#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc CSVs under <dir>/pmc_*/ into one JSON: per kernel family, the mean counter
value per launch, calibrated byte counts and the kernel's mean duration."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
GIB = float(1 << 30)


def family(name):
    if "csr_spmm_rowgroup_kernel" in name:
        return "csr_spmm_rowgroup_kernel"
    if "longrow_partial" in name:
        return "csr_spmm_longrow_partial_kernel"
    if "longrow_combine" in name:
        return "csr_spmm_longrow_combine_kernel"
    if "copyBuffer" in name:
        return "torch_copy_1GiB"
    if "reduce_kernel" in name:
        return "torch_sum_1GiB"
    return None


# counter -> family -> list of per-dispatch values (in dispatch order)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    meta = {}
    for row in csv.DictReader(open(path)):
        fam = family(row["Kernel_Name"])
        if fam is None:
            continue
        key = (row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        meta[row["Dispatch_Id"]] = (fam, int(row.get("Grid_Size", 0)))
    for (did, cname), v in sorted(per_dispatch.items(), key=lambda kv: int(kv[0][0])):
        fam, grid = meta[did]
        vals[cname][(fam, grid)].append(v)
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        fam = family(row["Kernel_Name"])
        if fam:
            durs[(fam, int(row.get("Grid_Size", 0)))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    break

out = {"counters_mean_per_launch": {}, "notes": []}
for cname, fams in vals.items():
    for (fam, grid), lst in fams.items():
        out["counters_mean_per_launch"].setdefault("%s[grid=%d]" % (fam, grid), {})[cname] = sum(lst) / len(lst)
for (fam, grid), lst in durs.items():
    out["counters_mean_per_launch"].setdefault("%s[grid=%d]" % (fam, grid), {})["duration_us_profiled"] = sum(lst) / len(lst)


def mean_of(counter, fam_prefix):
    for (fam, grid), lst in vals.get(counter, {}).items():
        if fam.startswith(fam_prefix):
            return sum(lst) / len(lst)
    return None


# calibration on the 1 GiB copy: known 2^30 bytes read and 2^30 written per launch
f_copy, w_copy = mean_of("FETCH_SIZE", "torch_copy"), mean_of("WRITE_SIZE", "torch_copy")
f_sum = mean_of("FETCH_SIZE", "torch_sum")
cal = {}
if f_copy:
    cal["fetch_bytes_per_unit"] = GIB / f_copy
if w_copy:
    cal["write_bytes_per_unit"] = GIB / w_copy
if f_sum:
    cal["fetch_bytes_per_unit_from_sum"] = GIB / f_sum
out["calibration"] = cal
out["notes"].append("FETCH_SIZE/WRITE_SIZE units are nominally KiB; bytes_per_unit is the measured factor on a "
                    "1 GiB copy (2048 would mean the documented 'reads half' behaviour of gfx950).")
spmm = {}
for key, c in out["counters_mean_per_launch"].items():
    if key.startswith("csr_spmm_rowgroup_kernel"):
        e = dict(c)
        if "FETCH_SIZE" in c and "fetch_bytes_per_unit" in cal:
            e["hbm_read_bytes_calibrated"] = c["FETCH_SIZE"] * cal["fetch_bytes_per_unit"]
        if "WRITE_SIZE" in c and "write_bytes_per_unit" in cal:
            e["hbm_write_bytes_calibrated"] = c["WRITE_SIZE"] * cal["write_bytes_per_unit"]
        if "hbm_read_bytes_calibrated" in e and "hbm_write_bytes_calibrated" in e:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_calibrated"] + e["hbm_write_bytes_calibrated"]
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        spmm[key] = e
out["csr_spmm"] = spmm
print(json.dumps(out, indent=1))
