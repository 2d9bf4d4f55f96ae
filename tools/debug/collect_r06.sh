# The round's evidence in ONE gpurun call (round 6): counter passes + kernel stats (tools/collect_pmc.sh), per-leg kernel stats and the
# N = 8192 loop's counters (tools/collect_legs.sh), then the two bench lines with the new summaries in place.
#   gpurun --timeout 3000 -- 'bash tools/debug/collect_r06.sh'
set -u
cd $GRAFT_REPO_ROOT
R=r06
O=gpurun_out/$R; mkdir -p $O
bash tools/collect_pmc.sh $O/pmc $R > $O/pmc.log 2>&1
bash tools/collect_legs.sh $O/legs $R > $O/legs.log 2>&1
cp $O/pmc/${R}_*.json $O/pmc/${R}_*.csv $O/legs/${R}_*.json $O/legs/${R}_*.csv $O/ 2>/dev/null
cp $O/${R}_pmc_*.json profiles/ 2>/dev/null            # bench.py reads them (guarded by source hashes)
python tools/debug/mx_check.py notest > $O/${R}_mx_check.json 2> $O/mx_check.log
python tools/debug/loss_in_scan_ab.py 32 64 > $O/${R}_loss_in_scan_ab.jsonl 2> $O/lis.log
python tools/scorer_time.py > $O/${R}_scorer.jsonl 2> $O/scorer.log
python tools/debug/sweep_small.py > $O/${R}_sweep_small.txt 2> $O/sweep.log
python bench.py > $O/${R}_bench_default.json 2> $O/bench_default.log
python bench.py --steps 20 --warmup 5 > $O/${R}_bench_k20.json 2> $O/bench_k20.log
cut -c1-300 $O/${R}_bench_default.json; echo; cut -c1-300 $O/${R}_bench_k20.json; echo
rm -rf $O/pmc $O/legs
ls -la $O | head -60
