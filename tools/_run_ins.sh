cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_stages.py tests/test_gpu_cabi_driver.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --repeats 7 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'], d['config']['capacity_mode']['equals_exact_path_bitwise'])"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python bench.py --no-cpu-baseline --no-extras --no-host-span --no-sharded --repeats 3 > /dev/null 2>&1
python tools/rocprof_summary.py "$(find /tmp/kt -name '*.db' | head -1)" gpurun_out/kstats_first.txt auto; grep -E "conv_first|insert_points" gpurun_out/kstats_first.txt | cut -c1-170
