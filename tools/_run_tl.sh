cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d /tmp/tl -- python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --steps 10 --warmup 3 > gpurun_out/tl_bench.log 2>&1
python tools/timeline.py "$(find /tmp/tl -name '*.db' | head -1)" full > gpurun_out/tl_timeline.txt 2>&1
tail -3 gpurun_out/tl_bench.log | cut -c1-300
