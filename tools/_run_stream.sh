cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
for cfg in "" "--f32" "--batch 1" "--sdma 0" "--batch 4"; do
echo "=== default env, cfg: $cfg"; timeout 300 python tools/stream_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -A22 "^SDMA" | cut -c1-200
done
timeout 900 python -m pytest tests/test_gpu_harness.py tests/test_gpu_emulation.py -x -q 2>&1 | tail -5
