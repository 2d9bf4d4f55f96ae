cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_emd.py tests/test_gpu_reference_checks.py tests/test_gpu_train.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/debug/emd_attack_ab.py 128 2>/dev/null | tail -3
timeout 300 python tools/debug/emd_attack_ab.py 32 2>/dev/null | tail -3
timeout 300 python tools/emd_attack_time.py 32 128 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['batch'], 'it', round(d['ms_per_iteration_chamfer_plus_emd'],3), 'match', round(d['approx_match_ms'],3), 'fused', round(d['fused_levels_cost_grad1_ms'],3), 'dense', {k: round(v,3) for k,v in d['every_sweep_dense'].items()})
"
