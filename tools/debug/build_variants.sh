#!/bin/bash
# Builds geometric_adv_amd/lib/variants/libgeoadv_<name>.so: the library with ONE translation unit recompiled with extra
# -D flags (run HERE, the files travel with the gpurun snapshot; delete the directory afterwards).
#   bash tools/debug/build_variants.sh chamfer_pk.hip base:-DPK_VARIANT=0 nores:-DPK_VARIANT=1 ...
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$1; shift
CS=$ROOT/geometric_adv_amd/csrc
OUT=$ROOT/geometric_adv_amd/lib/variants
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
make -C "$CS" -j8 > /dev/null
OTHERS=$(ls "$CS"/_obj/*.o | grep -v "/${SRC%.hip}.o")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c "$CS/$SRC" -o "/tmp/variant_$name.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgeoadv_$name.so" $OTHERS "/tmp/variant_$name.o"
  echo "built $OUT/libgeoadv_$name.so ($defs)"
done
