#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python tools/headline_host_profile.py > gpurun_out/headline_host.txt 2>&1
rm -rf gpurun_out/prof_hl
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_hl" -o hl -- python "$GRAFT_REPO_ROOT/tools/headline_host_profile.py" --trace) > gpurun_out/headline_trace.log 2>&1
python tools/headline_host_profile.py --analyse "$(find gpurun_out/prof_hl -name '*kernel_trace.csv' | head -1)" >> gpurun_out/headline_host.txt 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu --no-gat --no-trainer --no-shard-base --no-papers --no-pmc --no-sage 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --steps 20 --warmup 5: value', r['value'], 'ms_per_step', r['ms_per_step'], 'kernel_ms_in_step', r['roofline']['kernel_ms_in_step'])" >> gpurun_out/headline_host.txt 2>&1; done
head -60 gpurun_out/headline_host.txt | cut -c1-180
tail -25 gpurun_out/headline_host.txt | cut -c1-180
