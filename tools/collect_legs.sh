#!/bin/bash
# Per-leg evidence for bench.py's secondary.configs (run on the GPU box through gpurun): one `rocprofv3 --kernel-trace --stats`
# pass per leg -> <ROUND>_leg_<leg>_kernel_stats.csv + the leg's own JSON, and the FETCH_SIZE / WRITE_SIZE passes of the N = 8192
# loop's Chamfer kernels -> <ROUND>_pmc_chamfer_n8192.json (bench.py: config4_leg reads it, guarded by source hashes).
# Usage: tools/collect_legs.sh OUTDIR [ROUND]
set -u
OUT=${1:-gpurun_out/legs}
R=${2:-r06}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for leg in config2 config3 config4 trained_victim training; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/leg_$leg" -- python3 bench.py --only-leg $leg --steps 200 --warmup 20 > "$OUT/${R}_leg_$leg.json" 2> "$OUT/leg_$leg.err"
  rc=$?
  [ $rc -ne 0 ] && echo "leg $leg: rocprofv3 pass failed (rc $rc)" && tail -3 "$OUT/leg_$leg.err"
  f=$(ls "$OUT"/leg_$leg/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${R}_leg_${leg}_kernel_stats.csv"
done
export GEOADV_TOOL_N=8192
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/n8192_$i" -- python3 tools/attack_breakdown.py 32 > "$OUT/n8192_$i.log" 2>&1
  rc=$?
  [ $rc -ne 0 ] && echo "n8192 pmc pass $i failed (rc $rc)"
done
python3 tools/pmc_summary.py --hash chamfer_sym.hip,chamfer_mx.h,chamfer_grid.h "$OUT"/n8192_* > "$OUT/${R}_pmc_chamfer_n8192.json"
ls -la "$OUT"/*.csv "$OUT"/*.json
