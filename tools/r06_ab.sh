#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cpu_twins.py -q 2>&1 | tail -5
for lib in gpurun_ab/libimf_base.so ""; do
  IMF_LIB=${lib:+$PWD/$lib} VARIANT=3 timeout 600 python tools/sorted_conv_probe.py 2>&1 | grep -v amdgpu.ids
done
VARIANT=0 timeout 600 python tools/sorted_conv_probe.py 2>&1 | grep -v amdgpu.ids
