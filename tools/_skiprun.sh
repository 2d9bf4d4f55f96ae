cd /tmp && export TMPDIR=/tmp
for sk in 0; do
  GEOADV_TRAIN_SKIP=$sk timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$sk -- python3 /root/repo/tools/train_bench.py --steps 10 > /dev/null 2>&1
  f=$(find /tmp/ps_$sk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$sk" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "train_bwd_kernel" in r["Name"]]
print("skip=%s " % sys.argv[2] + "  ".join("%s: %.1f us" % (r["Name"].split("train_bwd_kernel")[1].split("(")[0], float(r["AverageNs"]) / 1e3) for r in sorted(rows, key=lambda r: r["Name"])))
PY
done
