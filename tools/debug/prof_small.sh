# kernel traces of the plain attack loop at several batch sizes: bash tools/debug/prof_small.sh OUTDIR B...
OUT=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rm -rf /tmp/pl
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pl -o pl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attack_breakdown.py $B > /tmp/pl.log 2>&1
  f=$(find /tmp/pl -name "*kernel_stats*" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/$OUT/loop_b${B}_kernel_stats.csv
  echo "== B=$B"
  python3 - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-62s calls %5s avg %8.2f us %6s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  grep -o '{"batch.*' /tmp/pl.log | cut -c1-400
done
