cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_grouping.py tests/test_gpu_defense.py -x -q -m gpu > gpurun_out/r4b/pytest1.log 2>&1; echo "pytest1 rc=$?" >> gpurun_out/r4b/pytest1.log
tail -5 gpurun_out/r4b/pytest1.log
timeout 300 python tools/defense_time.py > gpurun_out/r4b/defense.json 2> gpurun_out/r4b/defense.err; cat gpurun_out/r4b/defense.json; tail -3 gpurun_out/r4b/defense.err
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4b/bench_k20.json 2> gpurun_out/r4b/bench_k20.err; echo "bench rc=$?"; tail -5 gpurun_out/r4b/bench_k20.err; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4b/bench_k20.json"))
    print(d["value"], d["value_all_pairs"], d["roofline"]["frac"])
    print(json.dumps(d["secondary"]["configs"], indent=1)[:6000])
    print(json.dumps(d["secondary"]["trained_victim"], indent=1))
    print(d.get("paired_search"))
except Exception as e:
    print("parse failed", e)
PY
