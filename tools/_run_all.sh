cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6
