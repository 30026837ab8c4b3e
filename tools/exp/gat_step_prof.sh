#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_gat
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_gat" -o gat -- python "$GRAFT_REPO_ROOT/tools/gat_bench.py" --only fused-dropout --steps 20) > gpurun_out/prof_gat.log 2>&1
head -40 gpurun_out/prof_gat/gat_kernel_stats.csv | cut -c1-260
rm -f gpurun_out/prof_gat/gat_kernel_trace.csv
mkdir -p gpurun_out/mempmc2
i=0
for c in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf gpurun_out/mempmc2/p$i
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/mempmc2/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/mem_probe.py") > gpurun_out/mempmc2/p$i.log 2>&1
done
python tools/mem_summarize.py gpurun_out/mempmc2/p* > gpurun_out/mempmc2/summary.txt 2>&1
find gpurun_out/mempmc2 -name "*.csv" -size +2M -delete
grep -A8 "vrow_kernel<GatFwdOp\|vrow_kernel<SpmmOp\|vrow_kernel<GatBwd" gpurun_out/mempmc2/summary.txt | head -90
