#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_cpu_twins.py -q -x 2>&1 | tail -3
run() { echo "== v$1 sorted=$2"; IMF_SORTED_MAP=$2 IMF_CONV_VARIANT=$1 timeout 300 python tools/step_pair.py 2>&1 | grep -v amdgpu.ids | head -${3:-5}; }
run 3 0; run 3 7; run 3 1; run 3 0; run 3 7; run 3 3; run 3 4
run 0 0 2; run 0 7 2; run 6 0 2; run 6 7 2
