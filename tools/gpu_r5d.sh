#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
COGDL_AMD_ZOO_REPORT=$PWD/gpurun_out/r5d_zoo.json timeout 1500 python -m pytest tests/test_reference_zoo_gpu.py -q -p no:cacheprovider > gpurun_out/r5d_zoo.log 2>&1; echo "zoo rc=$?"; tail -12 gpurun_out/r5d_zoo.log | cut -c1-400
for t in "1=4096" "1=16384" "3=2040" "1=4096,3=2040"; do
  COGDL_AMD_TUNING=$t timeout 300 python tools/papers_bench.py --only symmetrised --steps 2 > gpurun_out/r5d_papers_$t.json 2> gpurun_out/r5d_papers_$t.err
  python -c "
import json,sys; r=json.load(open('gpurun_out/r5d_papers_$t.json'))['symmetrised']; print('$t', 'fwd', round(r['forward']['ms'],1), round(r['forward']['frac'],3), 'bwd', round(r['backward_alone']['ms'],1))"
done
