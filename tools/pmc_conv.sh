#!/bin/bash
# PMC passes over one sparse-conv shape: tools/pmc_conv.sh <variant> <split> [cin cout level]
# (counters in separate rocprofv3 runs, --kernel-trace only, as the pool requires)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/pmc_conv; rm -rf /tmp/pmc_conv; mkdir -p /tmp/pmc_conv $out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_conv/p$i -- python tools/conv_single.py "$@" > /tmp/pmc_conv/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 /tmp/pmc_conv/p$i.log)"
done
python tools/pmc_dump.py k_spconv "/tmp/pmc_conv/**/*.db" > $out/counters_v$1_s$2.txt 2>&1
cat $out/counters_v$1_s$2.txt
