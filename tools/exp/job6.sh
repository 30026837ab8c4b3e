#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/kt_step
(cd /tmp; rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_step -o kt -- python $GRAFT_REPO_ROOT/tools/gat_bench.py --leg --only fused-dropout --steps 10 > $GRAFT_REPO_ROOT/gpurun_out/kt_step.log 2>&1)
tail -3 gpurun_out/kt_step.log | cut -c1-300
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kt_step/kt_kernel_stats.csv')))
for r in rows[:40]:
    print("%-110s calls %4s total %9.2f ms avg %9.1f us" % (r['Name'].replace('void ','')[:110], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
