cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_grouping.py -x -q -m gpu 2>&1 | tail -5
bash tools/debug/ab_cmd.sh python tools/defense_time.py 2>/dev/null | tee gpurun_out/r4d/ab_knn.txt | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print({k: round(v, 4) for k, v in d.items() if 'knn' in k})
    else: print(ln.strip())
"
