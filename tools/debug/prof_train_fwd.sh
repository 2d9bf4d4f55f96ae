# per-kernel averages of the training step's layer kernels (one line per kernel): bash tools/debug/prof_train_fwd.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats -d /tmp/pt -o pt --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py --steps 30 > /tmp/pt.log 2>&1
f=$(find /tmp/pt -name "*kernel_stats*" | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r["Name"]
    if "train_fwd_kernel" in n or "train_bwd_fused" in n or "post_layer" in n or "bn_finalize" in n:
        out.append("%s %.1f" % (n.split("geoadv::")[1].split("(")[0], float(r["AverageNs"]) / 1e3))
print(" | ".join(sorted(out)))
PY
tail -1 /tmp/pt.log | cut -c1-120
