#!/bin/bash
# A/B of library variants on ONE box with an arbitrary command: every geometric_adv_amd/lib/variants/libgeoadv_*.so is swapped
# in, the command (arguments) runs, the original library is restored (also on failure).   bash tools/debug/ab_cmd.sh <command...>
set -eu
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cp geometric_adv_amd/lib/libgeoadv.so /tmp/libgeoadv_keep.so
trap 'cp /tmp/libgeoadv_keep.so "$GRAFT_REPO_ROOT/geometric_adv_amd/lib/libgeoadv.so"' EXIT
for v in geometric_adv_amd/lib/variants/libgeoadv_*.so; do
    cp "$v" geometric_adv_amd/lib/libgeoadv.so
    echo "==== $v"
    "$@" || echo "(command failed)"
done
