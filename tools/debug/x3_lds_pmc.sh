#!/bin/bash
# LDS-side counters of the encoder forward alone (B = 32, 300 launches) under either arithmetic -> OUT/<round>_pmc_encoder_lds.json
#   gpurun -- 'bash tools/debug/x3_lds_pmc.sh gpurun_out/r06 r06'
set -u
OUT=${1:-gpurun_out/r06}; R=${2:-r06}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in f16x2 bf16x3; do
  i=0
  for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    ARITH=$a timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "/tmp/x3lds_${a}_$i" -- python3 tools/debug/x3_time.py 32 100 > "/tmp/x3lds_${a}_$i.log" 2>&1 || { echo "$a pass $i FAILED"; tail -3 "/tmp/x3lds_${a}_$i.log"; rm -rf "/tmp/x3lds_${a}_$i"; }
  done
done
python3 tools/pmc_summary.py /tmp/x3lds_* > "$OUT/${R}_pmc_encoder_lds.json"
python3 - "$OUT/${R}_pmc_encoder_lds.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "encoder_fwd3" in k:
        print(k, {c: round(x["mean"] / 1e6, 3) for c, x in v.items() if isinstance(x, dict)}, v.get("avg_us_profiled"))
PY
