# kernel trace of tools/chamfer_pk_time.py: bash tools/debug/prof_pk.sh B...
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o pk --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/chamfer_pk_time.py "$@" > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*kernel_stats*" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-62s calls %5s avg %8.2f us min %8.2f max %8.2f" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat /tmp/pk.log | grep batch
