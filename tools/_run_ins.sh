cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" _ab/ins2.so _ab/ins4.so _ab/ins16.so; do
  if [ -n "$lib" ]; then export IMF_LIB=$PWD/$lib; else unset IMF_LIB; fi
  echo "== lib ${lib:-default(8)}"; python tools/insert_time.py 2>/dev/null | tail -1
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'])"
done
unset IMF_LIB; export IMF_INSERT_OLD=1; echo "== old"; timeout 300 python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'])"
