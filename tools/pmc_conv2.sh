#!/bin/bash
# second PMC battery (memory path) over one sparse-conv shape: tools/pmc_conv2.sh <variant> <split> [cin cout level]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/pmc_conv; rm -rf /tmp/pmc_conv2; mkdir -p /tmp/pmc_conv2 $out
i=0
for set in "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUBBLE_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_conv2/p$i -- python tools/conv_single.py "$@" > /tmp/pmc_conv2/p$i.log 2>&1 || echo "pass $i failed: $(grep -i -m2 "error\|invalid\|not" /tmp/pmc_conv2/p$i.log | cut -c1-200)"
done
python tools/pmc_dump.py k_spconv "/tmp/pmc_conv2/**/*.db" > $out/counters2_v$1_s$2.txt 2>&1
cat $out/counters2_v$1_s$2.txt
