#!/bin/bash
# Builds geometric_adv_amd/lib/variants/libgeoadv_<name>.so: the library with ONE translation unit recompiled with extra
# -D flags (run HERE, the files travel with the gpurun snapshot; delete the directory afterwards).
#   bash tools/debug/build_variants.sh chamfer_sym.hip base:-DX=0 other:-DX=1 ...
#   bash tools/debug/build_variants.sh all stamps:-DGA_STAMPS          (every translation unit recompiled)
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$1; shift
CS=$ROOT/geometric_adv_amd/csrc
OUT=$ROOT/geometric_adv_amd/lib/variants
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
make -C "$CS" -j8 > /dev/null
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  if [ "$SRC" = all ]; then
    objs=""
    for f in "$CS"/*.hip; do
      o="/tmp/variant_${name}_$(basename "${f%.hip}").o"
      /opt/rocm/bin/hipcc $FLAGS $defs -c "$f" -o "$o" &
      objs="$objs $o"
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgeoadv_$name.so" $objs
  else
    OTHERS=$(ls "$CS"/_obj/*.o | grep -v "/${SRC%.hip}.o")
    /opt/rocm/bin/hipcc $FLAGS $defs -c "$CS/$SRC" -o "/tmp/variant_$name.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgeoadv_$name.so" $OTHERS "/tmp/variant_$name.o"
  fi
  echo "built $OUT/libgeoadv_$name.so ($defs)"
done
