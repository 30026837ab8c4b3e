#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "13=0" "6=-2" "13=0" "6=-2"; do
  COGDL_AMD_TUNING=$t timeout 300 python tools/papers_bench.py --steps 2 > gpurun_out/r5p_papers_$t.json 2> gpurun_out/r5p_papers_$t.err
  python -c "
import json,sys; r=json.load(open('gpurun_out/r5p_papers_$t.json'))
for k in ('directed','symmetrised'):
    v=r[k]; print('$t', k, 'fwd', round(v['forward']['ms'],1), round(v['forward']['frac'],3), 'bwd', round(v['backward_alone']['ms'],1))"
done
