cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import emulate_3dmatch as E
E.make_dataset("/tmp/emu_c", 433, 0, 8)
PY
for cfg in "4 4" "8 2" "16 1" "16 2" "4 8" "2 8" "16 8" "6 3"; do
  set -- $cfg
  echo "== workers $1 npz_threads $2"; python -m imfnet_amd.generate_desc --source /tmp/emu_c/fragments --target /tmp/emu_c/d_$1_$2 --seeded_weights 0 --npz_level 1 --workers $1 --npz_threads $2 2>&1 | grep -o "wall.*"
  rm -rf /tmp/emu_c/d_$1_$2
done
