cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4f; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 tools/debug/knn_only.py > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py --hash grouping.hip $O/p* > $O/knn_pmc.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r4f/knn_pmc.json"))
for k,v in d.items():
    if "knn_fast_kernel" in k or "knn_grid_kernel" in k:
        print(k[:60], {c:(round(x["mean"]) if isinstance(x,dict) else round(x,1)) for c,x in v.items()})
PY
