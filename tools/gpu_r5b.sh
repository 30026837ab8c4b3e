#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bigcsr_gpu.py tests/test_config5_full_gpu.py "tests/test_ops_gpu.py::test_scatter_max_and_mhspmm_against_the_reference_cuda_kernels_golden" -q -p no:cacheprovider --durations=5 > gpurun_out/r5b_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r5b_tests.log
timeout 600 python tools/papers_bench.py > gpurun_out/r5b_papers_full.json 2> gpurun_out/r5b_papers_full.err; echo "full rc=$?"; cat gpurun_out/r5b_papers_full.json; tail -5 gpurun_out/r5b_papers_full.err
( time timeout 900 python bench.py > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err ) 2>&1 | tail -3; python -c "
import json; r=json.load(open('gpurun_out/r5b_bench.json')); print(r['value'], r['ms_per_step']); print(json.dumps(r['roofline'], indent=0)[:1500]); print(json.dumps(r.get('configs4_papers_1gpu'))[:600])"; tail -3 gpurun_out/r5b_bench.err
