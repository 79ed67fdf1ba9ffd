#!/bin/bash
# kernel trace of the graph-replay steps only; timeline of the last step
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=${1:-g}; mode=${2:-graph}; out=gpurun_out/r2; mkdir -p $out; rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
ONLY=$mode timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/kt -- python tools/graph_probe.py > $out/${tag}_probe.txt 2>/tmp/prof_$tag/kt.err
python tools/rocprof_summary.py "$(find /tmp/prof_$tag/kt -name '*.db' | head -1)" $out/${tag}_kernel_stats.txt 34
