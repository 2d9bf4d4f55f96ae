#!/bin/bash
# usage (on the GPU box, through gpurun): tools/debug/variant_bench.sh OUT v1 v2 ...   -- bench.py (headline only) under each library
# variant in _variants/ (copies of tools/debug/build_variants.sh's output: geometric_adv_amd/lib/variants is not in the snapshot path)
out=$1; shift
cp geometric_adv_amd/lib/libgeoadv.so /tmp/base_lib.so
trap 'cp /tmp/base_lib.so geometric_adv_amd/lib/libgeoadv.so' EXIT        # (the product library is put back whatever happens)
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base_lib.so geometric_adv_amd/lib/libgeoadv.so; else cp _variants/libgeoadv_$v.so geometric_adv_amd/lib/libgeoadv.so; fi
  python bench.py --steps 20 --warmup 5 --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), d['kernel_ms_per_iteration'])" | tee -a $out
done
cp /tmp/base_lib.so geometric_adv_amd/lib/libgeoadv.so
