#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_spmm_gpu.py tests/test_bigcsr_gpu.py tests/test_config5_full_gpu.py tests/test_config5_gpu.py tests/test_fused_gpu.py tests/test_dist_gpu.py -q -x -p no:cacheprovider > gpurun_out/r5o_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r5o_tests.log
timeout 600 python tools/papers_bench.py > gpurun_out/r5o_papers_full.json 2> gpurun_out/r5o_papers_full.err; python -c "
import json; r=json.load(open('gpurun_out/r5o_papers_full.json'))
for k in ('directed','symmetrised'):
    v=r[k]; print(k, 'fwd', round(v['forward']['ms'],1), round(v['forward']['frac'],3), 'bwd', round(v['backward_alone']['ms'],1), round(v['backward_alone']['frac'],3), 'fwd+bwd', round(v['forward_backward']['ms'],1), round(v['forward_backward']['frac'],3))
print(r['roofs'])"
timeout 300 python tools/papers_variants.py 2>&1 | head -4
