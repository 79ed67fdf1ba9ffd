#!/bin/bash
# tools/bench_ab.sh "<env assignments of arm 1>" "<arm 2>" ...: the headline leg of bench.py (pipelined pair step, no extras) per arm,
# alternating ROUNDS times on ONE box.  An arm is a string of VAR=value pairs ("" = the defaults), e.g.
#   tools/bench_ab.sh "IMF_L0_TAG=72 IMF_HALF_OCC4=0" ""        (round 5's half tiles everywhere vs the defaults)
for i in ${ROUNDS:-1 2 3}; do
  for arm in "$@"; do
    line=$(env $arm timeout 600 python bench.py --no-extras --no-cpu-baseline --no-host-span --no-sharded --full-out /tmp/bench_ab.json 2>/dev/null | tail -1)
    echo "[${arm:-defaults}] $(python -c "import json,sys; r=json.loads(sys.argv[1]); print(r['ms_per_step'], r['value'], r['roofline'].get('avg_launch_us'))" "$line")"
  done
done
