cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/km
GEOADV_KNN_MODE=grid rocprofv3 --kernel-trace --stats -d /tmp/km -o km --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/debug/knn_only.py > /tmp/km.log 2>&1
f=$(find /tmp/km -name "*kernel_stats*" | head -1)
python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_grid_build" in r["Name"]: print("build avg us", float(r["AverageNs"]) / 1e3)
PY
