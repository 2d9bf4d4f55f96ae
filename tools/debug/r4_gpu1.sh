cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_defense.py tests/test_gpu_ae_surface.py tests/test_gpu_next_rows.py -x -q -m gpu > gpurun_out/r4a/pytest1.log 2>&1; echo "pytest1 rc=$?" >> gpurun_out/r4a/pytest1.log
tail -15 gpurun_out/r4a/pytest1.log
timeout 300 python tools/defense_time.py > gpurun_out/r4a/defense.json 2> gpurun_out/r4a/defense.err; cat gpurun_out/r4a/defense.json; tail -3 gpurun_out/r4a/defense.err
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_grouping.py --deselect tests/test_gpu_defense.py --deselect tests/test_gpu_ae_surface.py --deselect tests/test_gpu_next_rows.py > gpurun_out/r4a/pytest2.log 2>&1; echo "pytest2 rc=$?" >> gpurun_out/r4a/pytest2.log
tail -8 gpurun_out/r4a/pytest2.log
