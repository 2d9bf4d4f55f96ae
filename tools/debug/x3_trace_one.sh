#!/bin/bash
# One kernel trace of the encoder forward alone (B = 32, 300 launches): prints the average duration of the encoder kernel.
# For ab_cmd.sh:  bash tools/debug/ab_cmd.sh bash tools/debug/x3_trace_one.sh
export TMPDIR=/tmp
rm -rf /tmp/x3t
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/x3t -o t --output-format csv -- python3 "$GRAFT_REPO_ROOT/tools/debug/x3_time.py" ${1:-32} ${2:-300} > /tmp/x3t.log 2>&1) || tail -5 /tmp/x3t.log
f=$(find /tmp/x3t -name "*kernel_stats*" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "encoder_fwd3" in r["Name"]:
        print("%-44s calls %s avg %.2f us min %.2f max %.2f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
