# The round's evidence on the final build, one gpurun call: full GPU test-suite, bench lines (driver form and default), tool outputs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > $O/r04_bench_k20.json 2> $O/bench_k20.err; echo "bench k20 rc=$? wall ${SECONDS}s"
timeout 1200 python bench.py > $O/r04_bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
timeout 600 python tools/defense_time.py > $O/r04_defense_b256.json 2>/dev/null
timeout 600 python tools/emd_attack_time.py 8 32 128 > $O/r04_emd_times.jsonl 2>/dev/null
timeout 900 python tools/attack_sweep.py > $O/r04_attack_sweep.json 2>/dev/null
timeout 600 python tools/scorer_time.py > $O/r04_scorer.jsonl 2>/dev/null
timeout 600 python tools/train_bench.py > $O/r04_train_bench.json 2>/dev/null
timeout 300 python tools/feed_probe.py > $O/r04_feed_probe.jsonl 2>/dev/null
python3 - <<'PY'
import json
for f in ("r04_bench_k20.json", "r04_bench_default.json"):
    d = json.load(open("gpurun_out/final/" + f))
    print(f, d["value"], d["value_all_pairs"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
cat $O/r04_defense_b256.json; cat $O/r04_train_bench.json
for B in 32 128; do
  rm -rf $O/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/debug/emd_trace.py run $B > /dev/null 2>&1
  python3 tools/debug/emd_trace.py show $O/tr > $O/r04_emd_trace_b$B.txt
  rm -rf $O/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/debug/emd_trace.py run $B blob > /dev/null 2>&1
  python3 tools/debug/emd_trace.py show $O/tr > $O/r04_emd_trace_blob_b$B.txt
done
tail -qn 1 $O/r04_emd_trace_b32.txt $O/r04_emd_trace_blob_b32.txt
