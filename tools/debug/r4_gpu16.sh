cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/g16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_emd.py tests/test_gpu_configs.py tests/test_gpu_grouping.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/emd_attack_time.py 32 128 > $O/emd_times.jsonl 2>/dev/null; cat $O/emd_times.jsonl | cut -c1-400
