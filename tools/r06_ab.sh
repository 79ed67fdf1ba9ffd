#!/bin/bash
# sorted decoder twins for the other arithmetics, with the sorts on the image stream (bench value = pipelined steps)
for v in 6 0; do for sm in 0 7 0 7; do
  echo "== variant $v IMF_SORTED_MAP=$sm"
  IMF_CONV_VARIANT=$v IMF_SORTED_MAP=$sm python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sharded --no-host-span --full-out /tmp/f.json 2>/dev/null | python -c "
import sys, json
c = json.loads(sys.stdin.read().splitlines()[-1])
print('  value ms/step', c['ms_per_step'], c['config']['issue'], c['config']['probe_ms_per_step'])"
done; done
