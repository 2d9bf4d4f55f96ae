cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_emd.py -x -q -m gpu 2>&1 | tail -12
