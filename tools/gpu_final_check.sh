#!/bin/bash
# last look at the tree as committed: smoke + the test files touched after the round's final full-suite run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_big_dispatch_gpu.py tests/test_bigcsr_gpu.py tests/test_sample_gpu.py tests/test_spmm_gpu.py tests/test_config5_full_gpu.py -q -p no:cacheprovider 2>&1 | tail -3
