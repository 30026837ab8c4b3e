#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "edge_softmax" > gpurun_out/r5h_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r5h_tests.log
timeout 600 python tools/es_split_ab.py > gpurun_out/r5h_es_split_ab.txt 2>&1; cat gpurun_out/r5h_es_split_ab.txt | tail -12
timeout 600 python -m pytest tests/test_config3_gpu.py tests/test_layout_independence_gpu.py -q -x -p no:cacheprovider > gpurun_out/r5h_tests2.log 2>&1; echo "tests2 rc=$?"; tail -5 gpurun_out/r5h_tests2.log
