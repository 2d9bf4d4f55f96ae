cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4g gpurun_out/r4h
timeout 1500 python -m pytest tests/test_gpu_emd.py tests/test_gpu_reference_checks.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python tools/emd_attack_time.py 32 128 2>/dev/null | tee gpurun_out/r4g/emd_times.jsonl
rm -rf gpurun_out/r4h/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4h/tr -- python3 tools/debug/emd_trace.py run > /dev/null 2>gpurun_out/r4h/err.txt
python3 tools/debug/emd_trace.py show gpurun_out/r4h/tr | head -12
