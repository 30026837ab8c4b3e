#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc counter_collection.csv into one line per dispatch of the row-reduce kernels.
usage: sq_summarize.py <dir>"""
import collections
import csv
import glob
import os
import re
import sys

for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    by = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        d = by.setdefault(int(row["Dispatch_Id"]), {"name": row["Kernel_Name"], "c": collections.defaultdict(float),
                                                    "t": (int(row["Start_Timestamp"]), int(row["End_Timestamp"]))})
        d["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    for did, d in by.items():
        if "rowreduce_main" not in d["name"]:
            continue
        short = re.sub(r"\(.*", "", d["name"]).replace("void cogdl::", "").replace("cogdl::", "")
        c = d["c"]
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        print("%-70s %8.1f us  " % (short[:70], (d["t"][1] - d["t"][0]) / 1e3) +
              "  ".join("%s %.3g (%.0f%%)" % (k.replace("SQ_", ""), v, 100 * v / wc) for k, v in sorted(c.items())))
