# kernel trace of the AE training step (batch 50 x 2048): bash tools/debug/prof_train.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats -d /tmp/pt -o pt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --steps 30 > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_stats*" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows[:40]:
    per_step = float(r["TotalDurationNs"]) / 35 / 1e3
    tot += per_step
    print("%-70s calls/step %5.1f avg %8.2f us  per step %8.2f us" % (r["Name"][:70], int(r["Calls"]) / 35, float(r["AverageNs"]) / 1e3, per_step))
print("sum per step us", tot)
PY
tail -1 /tmp/pt.log | cut -c1-300
