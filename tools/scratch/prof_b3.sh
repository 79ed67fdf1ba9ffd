cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
export IMF_CONV_VARIANT=${1:-3}
tag=r05v${IMF_CONV_VARIANT}
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/kt -- python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded > gpurun_out/r5/${tag}_bench_under_rocprof.json 2>/tmp/prof_$tag/kt.err || tail -5 /tmp/prof_$tag/kt.err
python tools/rocprof_summary.py "$(find /tmp/prof_$tag/kt -name '*.db' | head -1)" gpurun_out/r5/${tag}_kernel_stats.txt auto
bash tools/pmc_kernel.sh gpurun_out/r5/${tag}_pmc_raw.txt "k_spconv|k_fusion|k_pointwise" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1
python tools/pmc_table.py gpurun_out/r5/${tag}_pmc_raw.txt > gpurun_out/r5/${tag}_pmc_table.txt 2>&1
head -40 gpurun_out/r5/${tag}_kernel_stats.txt | cut -c1-150
cat gpurun_out/r5/${tag}_pmc_table.txt | cut -c1-200
