cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_harness.py tests/test_gpu_emulation.py -x -q 2>&1 | tail -3
timeout 900 python tools/cli_throughput.py 96 > gpurun_out/r4/cli_throughput.json 2> gpurun_out/r4/cli_throughput.err; echo rc=$?; tail -3 gpurun_out/r4/cli_throughput.err; python -c "
import json; d=json.load(open('gpurun_out/r4/cli_throughput.json'))
for k,v in d.items(): print(k, v)"
