#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spmm_gpu.py tests/test_reference_zoo_gpu.py -q -p no:cacheprovider > gpurun_out/r5l_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r5l_tests.log | cut -c1-300
