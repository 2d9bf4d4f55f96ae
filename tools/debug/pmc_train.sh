#!/bin/bash
# Counter passes of the AE training step (batch 50 x 2048): bash tools/debug/pmc_train.sh OUTDIR    (on the GPU box, through gpurun)
set -u
OUT=${1:-gpurun_out/pmc_train}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 tools/train_bench.py --steps 10 --warmup 2"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/train_$i" -- $CMD > "$OUT/train_$i.log" 2>&1
done
python3 tools/pmc_summary.py --hash train.hip,mfma_tile.h "$OUT"/train_* > "$OUT/pmc_train.json"
python3 - "$OUT/pmc_train.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "train_bwd_fused" in k or "train_fwd_kernel" in k or "post_layer" in k or "bn_finalize" in k or "dec_out_bwd" in k or "fc_out_fwd" in k:
        print(k[:60], {c: (round(x["mean"]) if isinstance(x, dict) else round(x, 1)) for c, x in v.items()})
PY
