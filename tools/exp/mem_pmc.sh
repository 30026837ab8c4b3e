#!/bin/bash
# counters of the memory pipeline over tools/mem_probe.py, one counter set per pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/mempmc
(cd /tmp && rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/mempmc/counters_list.txt" 2>&1)
i=0
for c in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD" \
         "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_TAG_STALL_sum TCC_READ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/mempmc/p$i
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -f csv -d "$GRAFT_REPO_ROOT/gpurun_out/mempmc/p$i" -o pmc -- python "$GRAFT_REPO_ROOT/tools/mem_probe.py") > gpurun_out/mempmc/p$i.log 2>&1
  echo "pass $i ($c) rc=$? $(tail -1 gpurun_out/mempmc/p$i.log | cut -c1-150)"
done
python tools/mem_summarize.py gpurun_out/mempmc/p* > gpurun_out/mempmc/summary.txt 2>&1
find gpurun_out/mempmc -name "*.csv" -size +2M -delete
cat gpurun_out/mempmc/summary.txt | head -150
