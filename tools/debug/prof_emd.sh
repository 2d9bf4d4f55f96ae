# kernel trace of tools/emd_attack_time.py B: bash tools/debug/prof_emd.sh B
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pe -o pe --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emd_attack_time.py ${1:-32} > /tmp/pe.log 2>&1
f=$(find /tmp/pe -name "*kernel_stats*" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-100s calls %5s avg %8.2f us %6s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
