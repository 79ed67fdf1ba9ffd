cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "--batch 2 --n 120" "--batch 1 --n 60" "--f32 --n 60"; do
echo "=== cfg: $cfg"; timeout 300 python tools/stream_probe.py $cfg 2>&1 | grep -v amdgpu.ids | grep -E "^stream|per job|sync|legs" | cut -c1-200
done
