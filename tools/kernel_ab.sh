#!/bin/bash
# A/B of one kernel under rocprofv3 --kernel-trace: tools/kernel_ab.sh <kernel-name-regex> "<ENV=1 ...>" ["<ENV2=1>" ...]
# prints the per-symbol averages of the matching kernels in the default bench, first without and then with each
# environment setting.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
pat=$1; shift
mkdir -p gpurun_out
for envs in "" "$@"; do
  rm -rf /tmp/kab
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kab -- python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('env[$envs] ms_per_step under rocprof', d['ms_per_step'])"
  python tools/rocprof_summary.py "$(find /tmp/kab -name '*.db' | head -1)" /tmp/kab_stats.txt 40 > /dev/null
  grep -E "$pat" /tmp/kab_stats.txt | head -4 | cut -c1-150
done
