#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_xcd_gpu.py -x -q 2>&1 | tail -5
python tools/xcd_quick.py auto 2>&1 | grep -v "^/opt" | grep "H=8\|csr_spmm" | tee gpurun_out/xcd_quick_combine.txt
rm -rf gpurun_out/kt
(cd /tmp; rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/kt -o kt -- python $GRAFT_REPO_ROOT/tools/gat_trace.py 8x8 > /dev/null 2>&1)
head -12 gpurun_out/kt/*kernel_stats.csv | cut -c1-150
