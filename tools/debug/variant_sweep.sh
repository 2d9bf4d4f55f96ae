#!/bin/bash
# usage (on the GPU box, through gpurun): tools/debug/variant_sweep.sh v1 v2 ...  -- the bench line's batch sweep under each library variant in _variants/
cp geometric_adv_amd/lib/libgeoadv.so /tmp/base_lib.so
trap 'cp /tmp/base_lib.so geometric_adv_amd/lib/libgeoadv.so' EXIT
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base_lib.so geometric_adv_amd/lib/libgeoadv.so; else cp _variants/libgeoadv_$v.so geometric_adv_amd/lib/libgeoadv.so; fi
  python bench.py --steps 100 --warmup 10 --windows 5 --no-cpu-baseline --no-rccl-selftest 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), {k:round(x,4) for k,x in d['strong_scaling']['measured_ms'].items()})"
done
