#!/bin/bash
# Round-6 artefacts on the GPU box, per arithmetic (bf16x3 = the default, f16x2 = fast mode, f32 = fp32 MFMA): kernel stats, the
# three HBM-traffic PMC passes, the SQ counter table -- then the default bench line, which reads those summaries back.
# usage: tools/round_profile_r06.sh [arith ...]   (outputs under gpurun_out/r6/; copy r06_* into profiles/)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/r6; mkdir -p $out
declare -A VAR=([bf16x3]=3 [f16x2]=6 [f32]=0)
for arith in ${@:-bf16x3 f16x2 f32}; do
  export IMF_CONV_VARIANT=${VAR[$arith]}
  tag=r06_$arith
  bash tools/profile_round.sh $tag > $out/profile_round_$arith.log 2>&1
  mv gpurun_out/${tag}_kernel_stats.txt $out/r06_kernel_stats_$arith.txt
  mv gpurun_out/${tag}_pmc_traffic.json $out/r06_pmc_traffic_$arith.json
  mv gpurun_out/${tag}_bench_under_rocprof.json $out/r06_bench_under_rocprof_$arith.json
  bash tools/pmc_kernel.sh $out/r06_pmc_raw_$arith.txt "k_" \
    "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
    "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1
  python tools/pmc_table.py $out/r06_pmc_raw_$arith.txt > $out/r06_pmc_counters_$arith.txt 2>&1
  rm -f $out/r06_pmc_raw_$arith.txt
  cp $out/r06_kernel_stats_$arith.txt $out/r06_pmc_traffic_$arith.json $out/r06_pmc_counters_$arith.txt profiles/
done
unset IMF_CONV_VARIANT
timeout 1200 python bench.py > $out/r06_bench.json 2> $out/r06_bench.err; echo "bench rc=$?"; tail -c 600 $out/r06_bench.json
ls -la $out
