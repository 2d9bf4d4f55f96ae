cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4e
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e/prof -- python3 tools/defense_time.py > gpurun_out/r4e/defense.json 2> gpurun_out/r4e/err.txt
f=$(ls gpurun_out/r4e/prof/*/*_kernel_stats.csv | head -1); cp "$f" gpurun_out/r4e/defense_kernel_stats.csv
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4e/defense_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("knn","outlier","critical","query_ball","encoder_fwd","chamfer","latent","fc2")):
        print(n[:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
