cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "--batch 2 --n 240" "--batch 4 --n 240" "--batch auto --n 240" "--batch 8 --n 240"; do
echo "=== cfg: $cfg"; timeout 300 python tools/stream_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -E "^stream|per job" | cut -c1-130
done
