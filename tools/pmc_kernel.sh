#!/bin/bash
# Hardware counters of selected kernels in the default bench, one rocprofv3 --pmc pass per quoted group
# (--kernel-trace only, as the pool requires).  usage: tools/pmc_kernel.sh <out.txt> <kernel regex> "<C1 C2 ..>" ["<..>" ...]
# Environment variables of the caller reach the bench (A/B switches).
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=$1; pat=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmck_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmck_$i -- python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --steps 10 --warmup 3 > /dev/null 2>/tmp/pmck_$i.err \
    || { echo "pass [$grp] failed: $(tail -2 /tmp/pmck_$i.err)" >> "$out"; continue; }
  python tools/pmc_dump.py "$(find /tmp/pmck_$i -name '*.db' | head -1)" "$pat" >> "$out"
done
cat "$out"
