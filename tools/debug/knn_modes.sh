# per-kernel averages of knn_dists(k = 8) at 256 x 2048 in the grid search's two forms: bash tools/debug/knn_modes.sh
cd /tmp && export TMPDIR=/tmp
for mode in grid grid_shells; do
  rm -rf /tmp/km
  GEOADV_KNN_MODE=$mode rocprofv3 --kernel-trace --stats -d /tmp/km -o km --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/debug/knn_only.py > /tmp/km.log 2>&1
  f=$(find /tmp/km -name "*kernel_stats*" | head -1)
  echo "== $mode"; python3 - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "knn" in r["Name"]: print("  %-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
