# like ab_variants.sh, for a given batch: prints the latent/grid launch and ms per iteration.  bash tools/debug/ab_variants_b.sh B [reps]
set -eu
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cp geometric_adv_amd/lib/libgeoadv.so /tmp/libgeoadv_keep.so
trap 'cp /tmp/libgeoadv_keep.so "$GRAFT_REPO_ROOT/geometric_adv_amd/lib/libgeoadv.so"' EXIT   # also on failure / interrupt
export TMPDIR=/tmp
for rep in $(seq 1 ${2:-2}); do
for v in geometric_adv_amd/lib/variants/libgeoadv_*.so; do
    cp $v geometric_adv_amd/lib/libgeoadv.so
    echo "== $v B=$1"
    bash tools/debug/prof_loop.sh $1 | { grep "latent\|scan_kernel" || true; }
    python3 -c "
import sys; sys.path.insert(0, 'tools'); import attack_sweep as s; print('ms_per_iteration', round(s.run($1, 2048, 300)['ms_per_iteration'], 4))" 2>/dev/null
done
done
