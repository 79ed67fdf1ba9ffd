cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_range.py -x -q 2>&1 | tail -4
