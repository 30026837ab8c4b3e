#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_fp16
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_fp16" -o ep -- python "$GRAFT_REPO_ROOT/tools/trainer_epoch.py" gpu 30 linear fp16) > gpurun_out/prof_fp16.log 2>&1
rm -f gpurun_out/prof_fp16/ep_kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_fp16/ep_kernel_stats.csv')))
tot=0
for r in rows:
    c=int(r['Calls'])
    if c % 30 == 0 or c % 31 == 0 or c % 32 == 0:
        per=float(r['TotalDurationNs'])/30/1e3
        tot+=per
        if per>8: print("%-130s %5.1f/epoch %8.1f us/epoch"%(r['Name'][:130], c/30, per))
print("sum: %.1f us/epoch"%tot)
PY
grep TRAINER gpurun_out/prof_fp16.log | cut -c1-200
