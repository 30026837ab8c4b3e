#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests/test_xcd_gpu.py tests/test_dist_gpu.py tests/test_reference_zoo_gpu.py -x -q 2>&1 | tail -15
python tools/xcd_quick.py auto 2>&1 | grep -v "^/opt" | grep "H=8" | tee gpurun_out/xcd_quick_combine2.txt
