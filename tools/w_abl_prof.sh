#!/bin/bash
# Kernel durations (rocprofv3 --kernel-trace) of the wave-split conv under each ablation library of tools/w_ablations.sh,
# per launch shape.   usage (GPU box): [BATCH=2] tools/w_abl_prof.sh "0 1 3 5 7 8" [staging]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for m in $1; do
  rm -rf /tmp/wab; echo "== mask $m"
  IMF_LIB=$PWD/imfnet_amd/_abl/libw_$m.so timeout 300 rocprofv3 --kernel-trace -d /tmp/wab -- python tools/conv_iso.py ${2:-wave8} > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/wab/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, grid_x/workgroup_x, grid_y, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%k_spconv_w%' group by 1,2,3 order by 2 desc, 3"):
    print("  %-44s grid=(%5d,%d) n=%3d avg %7.2f us  min %7.2f" % (r[0][:44], r[1], r[2], r[3], r[4], r[5]))
PY
done
