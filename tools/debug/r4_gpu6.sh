cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_grouping.py -x -q -m gpu 2>&1 | tail -5
bash tools/debug/ab_cmd.sh python tools/debug/knn_only.py 2>&1 | grep -v amdgpu.ids
