cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
rm -rf /tmp/kt; timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -- python tools/stream_probe.py --n 30 > gpurun_out/r4/stream_under_rocprof.txt 2>/tmp/kt.err
db=$(find /tmp/kt -name '*.db' | head -1); echo "db=$db"
python tools/stream_timeline.py $db gpurun_out/r4/stream_timeline_5streams.txt; cat gpurun_out/r4/stream_timeline_5streams.txt | head -150
