#!/bin/bash
# HBM-side traffic of csr_spmm at papers100M scale: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over
# tools/papers_bench.py --only directed --steps 1 (its roofs probe starts with the 1 GiB copies the units are calibrated on).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcpapers_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcpapers_$c" -o pmc -- python "$GRAFT_REPO_ROOT/tools/papers_bench.py" --only directed --steps 1) > gpurun_out/pmcpapers_$c.log 2>&1
  echo "pass $c rc=$?"; tail -2 gpurun_out/pmcpapers_$c.log | cut -c1-300
done
python tools/pmc_by_kernel.py gpurun_out pmcpapers_ SpmmOp > gpurun_out/pmcpapers_summary.json 2> gpurun_out/pmcpapers_summary.err
python -c "
import json; r=json.load(open('gpurun_out/pmcpapers_summary.json')); print(r['calibration'])
for k,v in r['kernels'].items():
    print(k[:100], v.get('launches'), round(v.get('duration_us_profiled',0)), 'us', round(v.get('hbm_bytes_per_launch',0)/1e9,1), 'GB per launch', round(v.get('hbm_GBs',0)), 'GB/s')"
rm -rf gpurun_out/pmcpapers_FETCH_SIZE gpurun_out/pmcpapers_WRITE_SIZE
