# per-launch traces of one fused approx-EMD call at B = 32 and 128 -> profiles/r04_emd_trace_b32.txt, _b128.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
for B in 32 128; do
  rm -rf $O/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/debug/emd_trace.py run $B > /dev/null 2>&1
  python3 tools/debug/emd_trace.py show $O/tr > $O/r04_emd_trace_b$B.txt; tail -1 $O/r04_emd_trace_b$B.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04_bench_k20_b.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/final/r04_bench_k20_b.json')); print(d['value'], d['secondary']['roofline_emd']['B32'])"
