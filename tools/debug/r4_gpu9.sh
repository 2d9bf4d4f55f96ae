cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_emd.py -x -q -m gpu 2>&1 | tail -3
for v in lv3 lv2; do
  cp geometric_adv_amd/lib/libgeoadv.so /tmp/keep.so; cp geometric_adv_amd/lib/variants/libgeoadv_$v.so geometric_adv_amd/lib/libgeoadv.so
  echo "== $v"; timeout 300 python tools/emd_attack_time.py 32 128 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['batch'], 'it', round(d['ms_per_iteration_chamfer_plus_emd'],3), 'match', round(d['approx_match_ms'],3), 'fused', round(d['fused_levels_cost_grad1_ms'],3), 'dense', d['every_sweep_dense'])
"
  for B in 32 128; do rm -rf gpurun_out/r4h/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4h/tr -- python3 tools/debug/emd_trace.py run $B > /dev/null 2>&1; python3 tools/debug/emd_trace.py show gpurun_out/r4h/tr | head -9 | awk '{printf "%s %s | ", $1, $(NF-1)} END {print ""}'; done
  cp /tmp/keep.so geometric_adv_amd/lib/libgeoadv.so
done
