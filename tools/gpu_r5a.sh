#!/bin/bash
# round 5, call A: the 64-bit CSR path (small tests, reference-kernel goldens, scaled + full-size papers leg, full-size test)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/golden
timeout 600 python -m pytest tests/test_bigcsr_gpu.py -x -q -p no:cacheprovider > gpurun_out/r5a_bigcsr_tests.log 2>&1; echo "bigcsr tests rc=$?"; tail -5 gpurun_out/r5a_bigcsr_tests.log
timeout 300 python tests/golden/make_golden_gpu.py gpurun_out/golden > gpurun_out/r5a_golden.log 2>&1; echo "golden rc=$?"; tail -3 gpurun_out/r5a_golden.log
COGDL_AMD_TUNING=15=60000000 timeout 300 python tools/papers_bench.py --nodes 20000000 --pairs 200000000 --steps 2 > gpurun_out/r5a_papers_scaled.json 2> gpurun_out/r5a_papers_scaled.err; echo "scaled rc=$?"; cat gpurun_out/r5a_papers_scaled.json; tail -3 gpurun_out/r5a_papers_scaled.err
timeout 600 python tools/papers_bench.py > gpurun_out/r5a_papers_full.json 2> gpurun_out/r5a_papers_full.err; echo "full rc=$?"; cat gpurun_out/r5a_papers_full.json; tail -5 gpurun_out/r5a_papers_full.err
timeout 900 python -m pytest tests/test_config5_full_gpu.py -x -q -p no:cacheprovider > gpurun_out/r5a_full_test.log 2>&1; echo "full test rc=$?"; tail -15 gpurun_out/r5a_full_test.log
