# A/B of library variants on ONE box: for every geometric_adv_amd/lib/variants/libgeoadv_*.so, swap it in, run the command
# given as arguments (default: kernel trace of the B = 32 loop), print the chamfer lines, restore.
#   bash tools/debug/ab_variants.sh [reps]
set -eu
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cp geometric_adv_amd/lib/libgeoadv.so /tmp/libgeoadv_keep.so
trap 'cp /tmp/libgeoadv_keep.so "$GRAFT_REPO_ROOT/geometric_adv_amd/lib/libgeoadv.so"' EXIT   # also on failure / interrupt
export TMPDIR=/tmp
for rep in $(seq 1 ${1:-2}); do
for v in geometric_adv_amd/lib/variants/libgeoadv_*.so; do
    cp $v geometric_adv_amd/lib/libgeoadv.so
    rm -rf /tmp/ab
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ab -o ab --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attack_breakdown.py 32 > /tmp/ab.log 2>&1)
    f=$(find /tmp/ab -name "*kernel_stats*" | head -1)
    echo "== $v"; grep -E "chamfer_sym|encoder_fwd2_kernel<true|loss_cgrad" $f | awk -F, '{printf "%s %s us\n", substr($1,1,48), $4/1000}'
    grep -o '"ms_per_iteration": [0-9.]*' /tmp/ab.log | head -1
done
done
