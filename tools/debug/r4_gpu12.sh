cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_attack.py tests/test_gpu_chamfer.py tests/test_gpu_chamfer_shapes.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
cp geometric_adv_amd/lib/libgeoadv.so /tmp/keep.so
for v in mc2 mc6 mc10 mc16; do
  cp geometric_adv_amd/lib/variants/libgeoadv_$v.so geometric_adv_amd/lib/libgeoadv.so
  echo "== $v"; timeout 300 python bench.py --only-leg trained_victim --steps 200 --warmup 20 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('search', round(d['grid_search']['attack_iterations_per_sec']), d['grid_search']['clouds_handed_back_of_32_along_the_attack'], 'all_pairs', round(d['all_pairs']['attack_iterations_per_sec']), 'adaptive', round(d['adaptive_default'].get('attack_iterations_per_sec', 0)), d['adaptive_default']['search_switched_off'], 'parity', d['index_parity_4_clouds'])
"
  timeout 300 python bench.py --steps 100 --warmup 10 --windows 3 --no-secondary --no-rccl-selftest --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['value']), 'all_pairs', round(d['value_all_pairs']), d['paired_search']['clouds_handed_back_of_32_at_the_end'])
"
done
cp /tmp/keep.so geometric_adv_amd/lib/libgeoadv.so
