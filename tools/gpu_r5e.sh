#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for t in "1=512" "1=256" "1=2048"; do
  COGDL_AMD_TUNING=$t timeout 300 python tools/papers_bench.py --only symmetrised --steps 2 > gpurun_out/r5e_papers_$t.json 2> gpurun_out/r5e_papers_$t.err
  python -c "
import json,sys; r=json.load(open('gpurun_out/r5e_papers_$t.json'))['symmetrised']; print('$t', 'fwd', round(r['forward']['ms'],1), round(r['forward']['frac'],3), 'bwd', round(r['backward_alone']['ms'],1))"
done
