#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/unroll16_ab.py > gpurun_out/r5i_unroll16_ab.txt 2>&1; tail -12 gpurun_out/r5i_unroll16_ab.txt
timeout 900 python -m pytest tests/test_spmm_gpu.py tests/test_fused_gpu.py tests/test_captured_step_gpu.py -q -x -p no:cacheprovider > gpurun_out/r5i_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5i_tests.log
