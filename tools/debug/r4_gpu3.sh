cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_defense.py -x -q -m gpu > gpurun_out/r4c/pytest1.log 2>&1; echo "pytest1 rc=$?" >> gpurun_out/r4c/pytest1.log
tail -25 gpurun_out/r4c/pytest1.log
timeout 300 python tools/defense_time.py > gpurun_out/r4c/defense.json 2> gpurun_out/r4c/defense.err; cat gpurun_out/r4c/defense.json; tail -3 gpurun_out/r4c/defense.err
