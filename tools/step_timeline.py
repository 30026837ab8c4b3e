#!/usr/bin/env python3
"""One replay of the captured mini-batch step out of a rocprofv3 --kernel-trace CSV: the kernels between two
consecutive first-hop sample_prep launches in the middle of the run, with durations and the gaps between them.
usage: step_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sample_prep_kernel" in r["Kernel_Name"] or "sample_counts_kernel" in r["Kernel_Name"]]
mid = len(idx) // 2
mid -= mid % 2  # (two hops per step)
i0, i1 = idx[mid], idx[mid + 2]
t0, prev = int(rows[i0]["Start_Timestamp"]), int(rows[i0]["Start_Timestamp"])
busy = 0.0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print("%8.1f us  dur %6.1f  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:100]))
    prev = e
print("nodes %d, span %.1f us, busy %.1f us" % (i1 - i0, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, busy))
