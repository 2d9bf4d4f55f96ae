#!/bin/bash
# PMC passes of the bench command and of the EMD timing tool (run on the GPU box through gpurun; counters in separate
# passes, --kernel-trace only beside --pmc: MI355X_MICROARCH.md / rocprofv3 PMC slots).  Usage: tools/collect_pmc.sh OUTDIR
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 40 --warmup 5 --windows 1 --no-cpu-baseline"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/bench_$i" -- $BENCH > "$OUT/bench_$i.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT"/bench_* > "$OUT/r02_pmc_encoder.json"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/emd_1" -- python3 tools/emd_attack_time.py 32 > "$OUT/emd_1.log" 2>&1
python3 tools/pmc_summary.py "$OUT"/emd_1 > "$OUT/r02_pmc_emd.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- $BENCH > "$OUT/bench_stats.log" 2>&1
cp "$OUT"/bench_stats/*/*_kernel_stats.csv "$OUT/r02_bench_kernel_stats.csv" 2>/dev/null
tail -2 "$OUT/bench_stats.log"
