#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/fwd16pmc
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
         "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_TAG_STALL_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/fwd16pmc/p$i
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/fwd16pmc/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/exp/fwd16_probe.py") > gpurun_out/fwd16pmc/p$i.log 2>&1
  echo "pass $i ($c) rc=$? $(tail -1 gpurun_out/fwd16pmc/p$i.log | cut -c1-120)"
done
MEM_SUMMARIZE_MATCH=linear_fwd16 python tools/mem_summarize.py gpurun_out/fwd16pmc/p* > gpurun_out/fwd16pmc/summary.txt 2>&1
find gpurun_out/fwd16pmc -name "*.csv" -size +1M -delete
cat gpurun_out/fwd16pmc/summary.txt | head -70
