(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -3)
python tools/rulebook_time.py 2>&1 | grep -v amdgpu | tail -6
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03t.json 2>/dev/null
python - <<P
import json
d=json.load(open("gpurun_out/r03t.json"))
print(d["ms_per_step"], d["timing"]["ms_per_step_min"], {k:v["avg_launch_us"] for k,v in d["roofline"]["per_kernel"].items()})
P
export TMPDIR=/tmp; rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python bench.py --no-cpu-baseline --no-extras --steps 10 --repeats 2 --settle-ms 100 > /dev/null 2>&1
python tools/rocprof_summary.py "$(find /tmp/kt -name '*.db' | head -1)" gpurun_out/r03t_kernel_stats.txt 30 > /dev/null; grep -E "k_rulebook" gpurun_out/r03t_kernel_stats.txt | head -12 | cut -c1-130
