cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/bp -o bp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attack_breakdown.py 32 > /dev/null 2>&1
f=$(find /tmp/bp -name "*kernel_stats*" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/loop_b32_kernel_stats.csv; head -12 $f | cut -c1-140
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/pm -o pm --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attack_breakdown.py 32 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pm > $GRAFT_REPO_ROOT/gpurun_out/pmc_chamfer_b32.json
python3 - <<PY
import json
d=json.load(open("$GRAFT_REPO_ROOT/gpurun_out/pmc_chamfer_b32.json"))
for k,v in d.items():
    if "chamfer_sym" in k: print(k[:50], {c:round(x["mean"]) for c,x in v.items() if isinstance(x,dict)})
PY
