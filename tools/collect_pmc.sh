#!/bin/bash
# The round's counter evidence (run on the GPU box through gpurun; counters in separate passes, --kernel-trace only beside
# --pmc: MI355X_MICROARCH.md / rocprofv3 PMC slots).  Usage: tools/collect_pmc.sh OUTDIR [ROUND]
#   <ROUND>_pmc_encoder.json      bench command: SQ issue counters, MFMA busy + GRBM, FETCH_SIZE, WRITE_SIZE (hashes encoder.hip, encoder_x3.hip, encoder_x3.h, mfma_tile.h)
#   <ROUND>_pmc_chamfer_hbm.json  tools/attack_breakdown.py 32 (the pruned loop, ONE leg): FETCH_SIZE, WRITE_SIZE, SQ issue counters
#                                 of chamfer_sym_kernel / loss_cgrad_kernel (hashes chamfer_sym.hip, chamfer_grid.h)
#   <ROUND>_pmc_emd.json          tools/emd_attack_time.py 32: SQ issue counters
#   <ROUND>_pmc_knn.json          tools/debug/knn_only.py (knn_dists at 256 x 2048, all-points and grid kernels): SQ counters, FETCH / WRITE
#   <ROUND>_bench_kernel_stats.csv, <ROUND>_loop_b32_kernel_stats.csv   --kernel-trace --stats of the two commands
set -u
OUT=${1:-gpurun_out/pmc}
R=${2:-r06}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the headline leg only: no secondary legs, no RCCL self-test child (a second profiled process on the GPU whose CSVs the summary
# would glob, inside the time limit of a counter-serialised pass)
BENCH="python3 bench.py --steps 40 --warmup 5 --windows 1 --no-cpu-baseline --no-secondary --no-rccl-selftest"
LOOP="python3 tools/attack_breakdown.py 32"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/bench_$i" -- $BENCH > "$OUT/bench_$i.log" 2>&1 || { echo "bench counter pass $i FAILED (rc $?): not summarised"; rm -rf "$OUT/bench_$i"; }
done
python3 tools/pmc_summary.py "$OUT"/bench_* > "$OUT/${R}_pmc_encoder.json"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/loop_$i" -- $LOOP > "$OUT/loop_$i.log" 2>&1 || { echo "loop counter pass $i FAILED (rc $?): not summarised"; rm -rf "$OUT/loop_$i"; }
done
python3 tools/pmc_summary.py --hash chamfer_sym.hip,chamfer_mx.h,chamfer_grid.h "$OUT"/loop_* > "$OUT/${R}_pmc_chamfer_hbm.json"
# the symmetric scan ALONE (operator form, no riders): the unscreened kernel at the loop's shape, the screened one at 2048 and 8192
for what in "sym unscreened 2048 chamfer_sym_alone" "mx screened 2048 chamfer_mx_alone" "mx8k screened 8192 chamfer_mx_n8192_alone"; do
  set -- $what
  i=0
  for cs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d "$OUT/$1_$i" -- python3 tools/debug/sym_only.py $2 $3 > "$OUT/$1_$i.log" 2>&1 || { echo "$1 counter pass $i FAILED (rc $?)"; rm -rf "$OUT/$1_$i"; }
  done
  python3 tools/pmc_summary.py --hash chamfer_sym.hip,chamfer_mx.h "$OUT"/$1_[0-9] > "$OUT/${R}_pmc_$4.json"
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/emd_1" -- python3 tools/emd_attack_time.py 32 > "$OUT/emd_1.log" 2>&1 || echo "emd counter pass FAILED (rc $?)"
python3 tools/pmc_summary.py --hash emd.hip "$OUT"/emd_1 > "$OUT/${R}_pmc_emd.json"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/knn_$i" -- python3 tools/debug/knn_only.py > "$OUT/knn_$i.log" 2>&1 || { echo "knn counter pass $i FAILED (rc $?)"; rm -rf "$OUT/knn_$i"; }
done
python3 tools/pmc_summary.py --hash grouping.hip "$OUT"/knn_* > "$OUT/${R}_pmc_knn.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- $BENCH > "$OUT/bench_stats.log" 2>&1
cp "$(grep -l encoder_fwd "$OUT"/bench_stats/*/*_kernel_stats.csv | head -1)" "$OUT/${R}_bench_kernel_stats.csv" 2>/dev/null   # (the RCCL self-test child writes a file of its own)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/loop_stats" -- $LOOP > "$OUT/loop_stats.log" 2>&1
cp "$(grep -l encoder_fwd "$OUT"/loop_stats/*/*_kernel_stats.csv | head -1)" "$OUT/${R}_loop_b32_kernel_stats.csv" 2>/dev/null
tail -2 "$OUT/bench_stats.log"
