cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash tools/round_profile_r04.sh 2>&1 | tail -30
