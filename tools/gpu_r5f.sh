#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_captured_step_gpu.py tests/test_sample_gpu.py tests/test_message_ops_gpu.py tests/test_bigcsr_gpu.py tests/test_pipeline_gpu.py -q -x -p no:cacheprovider > gpurun_out/r5f_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r5f_tests.log
timeout 600 python tools/thresh_sweep.py > gpurun_out/r5f_thresh_sweep.txt 2>&1; cat gpurun_out/r5f_thresh_sweep.txt | tail -20
timeout 300 python tools/csc_probe.py > gpurun_out/r5f_csc_probe.txt 2>&1; tail -30 gpurun_out/r5f_csc_probe.txt
timeout 300 python tools/sage_bench.py --captured --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-300
