#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files (one directory per pass) into one line per kernel NAME: the LAST dispatch
of each row-reduce kernel, its duration and every counter collected for it.  usage: mem_summarize.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import re
import sys

out = collections.OrderedDict()
for root in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        by = collections.OrderedDict()
        for row in csv.DictReader(open(path)):
            d = by.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                        "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
            d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
        last = collections.OrderedDict()
        for did, d in by.items():
            if os.environ.get("MEM_SUMMARIZE_MATCH", "rowreduce_") in d["name"]:
                last[d["name"]] = d
        for name, d in last.items():
            short = re.sub(r"\(.*", "", name).replace("void cogdl::", "").replace("cogdl::", "")
            o = out.setdefault(short, {"us": [], "c": collections.OrderedDict()})
            o["us"].append((d["t"][1] - d["t"][0]) / 1e3)
            o["c"].update(d["c"])
for short, o in out.items():
    print("%s" % short[:110])
    print("    %.1f us (passes: %s)" % (sum(o["us"]) / len(o["us"]), " ".join("%.0f" % u for u in o["us"])))
    for k, v in o["c"].items():
        print("    %-40s %.4g" % (k, v))
