#!/bin/bash
# Round-4 artefacts on the GPU box: bench line, kernel stats, HBM-traffic PMC passes, SQ counter table, stream timeline.
# usage: tools/round_profile_r04.sh   (outputs under gpurun_out/r4/; copy the summaries into profiles/)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/r4; mkdir -p $out
timeout 900 python bench.py > $out/r04_bench.json 2> $out/r04_bench.err; echo "bench rc=$?"
bash tools/profile_round.sh r04 > $out/profile_round.log 2>&1; mv gpurun_out/r04_* $out/ 2>/dev/null
bash tools/pmc_kernel.sh $out/r04_pmc_raw.txt "k_" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1
python tools/pmc_table.py $out/r04_pmc_raw.txt > $out/r04_pmc_table.txt 2>&1
# the host-array stream with copy-engine transfers: kernel + memory-copy trace of 30 jobs
rm -rf /tmp/kt_stream; timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt_stream -- python tools/stream_probe.py --n 30 > $out/r04_stream_probe.txt 2>/tmp/kt_stream.err
python tools/stream_timeline.py "$(find /tmp/kt_stream -name '*.db' | head -1)" $out/r04_stream_timeline.txt 2>&1 | tail -3
ls -la $out
