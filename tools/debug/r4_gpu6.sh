cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
true
cp geometric_adv_amd/lib/libgeoadv.so /tmp/keep.so
for v in t512 t1024; do
cp geometric_adv_amd/lib/variants/libgeoadv_$v.so geometric_adv_amd/lib/libgeoadv.so
rm -rf gpurun_out/r4e/prof2; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4e/prof2 -- python3 tools/debug/knn_only.py > /dev/null 2>&1
f=$(ls gpurun_out/r4e/prof2/*/*_kernel_stats.csv | head -1); python3 - "$f" $v <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "knn_grid" in r["Name"]: print(sys.argv[2], r["Name"][:45], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
cp /tmp/keep.so geometric_adv_amd/lib/libgeoadv.so
