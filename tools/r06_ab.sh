#!/bin/bash
python -m pytest tests/test_gpu_graph.py tests/test_gpu_harness.py -q -x 2>&1 | tail -3
for rep in 1 2; do for sm in 0 7; do
  echo "== IMF_SORTED_MAP=$sm"
  IMF_SORTED_MAP=$sm python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-sharded --full-out /tmp/f.json 2>/dev/null | python -c "
import sys, json
c = json.loads(sys.stdin.read().splitlines()[-1])
print('  value ms/step', c['ms_per_step'], 'issue', c['config']['issue'], c['config']['probe_ms_per_step'], 'host_span', c['host_span']['ms_per_step'], c['host_span']['vs_device_resident'])"
done; done
