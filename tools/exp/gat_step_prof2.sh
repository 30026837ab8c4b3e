#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_gat
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_gat" -o gat -- python "$GRAFT_REPO_ROOT/tools/gat_bench.py" --only "fused-dropout (attn_drop 0.5 = model default; install(fused_gat_dropout=True)) bf16" --steps 20) > gpurun_out/prof_gat.log 2>&1
rm -f gpurun_out/prof_gat/gat_kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_gat/gat_kernel_stats.csv')))
steps=22
tot=0
for r in rows:
    c=int(r['Calls'])
    if c % steps == 0 and c//steps <= 40:
        per=float(r['TotalDurationNs'])/steps/1e3
        tot+=per
        if per>15: print("%-120s %3d/step %8.1f us/step"%(r['Name'][:120], c//steps, per))
print("sum of kernels with calls divisible by %d: %.1f us/step"%(steps,tot))
PY
grep "ms per full" gpurun_out/prof_gat.log
