#!/bin/bash
# Round profile of the default bench command on the GPU box: kernel-trace stats + the three PMC passes
# of the HBM-traffic recipe (separate runs, --kernel-trace only).  Writes small summaries to gpurun_out/.
# usage: tools/profile_round.sh <tag>     (e.g. r01c)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-rXX}; out=gpurun_out; mkdir -p $out; rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
CMD="python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --steps 10 --warmup 3"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/kt -- python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded > $out/${tag}_bench_under_rocprof.json 2>/tmp/prof_$tag/kt.err
python tools/rocprof_summary.py "$(find /tmp/prof_$tag/kt -name '*.db' | head -1)" $out/${tag}_kernel_stats.txt auto
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_$tag/$n -- $CMD > /dev/null 2>/tmp/prof_$tag/$n.err || echo "pmc pass $n failed"
done
python tools/pmc_traffic.py "$(find /tmp/prof_$tag/FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/prof_$tag/WRITE_SIZE -name '*.db' | head -1)" \
       $out/${tag}_pmc_traffic.json "$(find /tmp/prof_$tag/TCC_HIT_sum -name '*.db' | head -1)"
head -14 $out/${tag}_kernel_stats.txt | cut -c1-160
