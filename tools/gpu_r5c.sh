#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bigcsr_gpu.py tests/test_config5_full_gpu.py tests/test_message_ops_gpu.py "tests/test_ops_gpu.py::test_scatter_max_and_mhspmm_against_the_reference_cuda_kernels_golden" -q -p no:cacheprovider --durations=5 > gpurun_out/r5c_tests.log 2>&1; echo "tests rc=$?"; grep -v "^  File\|^$" gpurun_out/r5c_tests.log | tail -40
COGDL_AMD_ZOO_REPORT=$PWD/gpurun_out/r5c_zoo.json timeout 1500 python -m pytest tests/test_reference_zoo_gpu.py -q -p no:cacheprovider > gpurun_out/r5c_zoo.log 2>&1; echo "zoo rc=$?"; tail -30 gpurun_out/r5c_zoo.log | cut -c1-600
timeout 600 python tools/papers_bench.py > gpurun_out/r5c_papers_full.json 2> gpurun_out/r5c_papers_full.err; echo "full rc=$?"; cat gpurun_out/r5c_papers_full.json; tail -5 gpurun_out/r5c_papers_full.err
