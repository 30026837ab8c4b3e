cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_lin$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_lin$i" -o p -- python "$GRAFT_REPO_ROOT/tools/linear_probe.py") > gpurun_out/pmc_lin$i.log 2>&1
  tail -2 gpurun_out/pmc_lin$i.log
done
