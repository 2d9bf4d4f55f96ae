# second half of a round's evidence: counters + the two bench lines (run through gpurun)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
bash tools/collect_pmc.sh $O/pmc r03 > $O/pmc.log 2>&1
cp $O/pmc/r03_pmc_encoder.json $O/pmc/r03_pmc_chamfer_hbm.json $O/pmc/r03_pmc_emd.json profiles/ 2>/dev/null
cp $O/pmc/r03_bench_kernel_stats.csv $O/pmc/r03_loop_b32_kernel_stats.csv $O/ 2>/dev/null
python bench.py > $O/r03_bench_default.json 2> $O/bench_default.log
python bench.py --steps 20 --warmup 5 > $O/r03_bench_k20.json 2> $O/bench_k20.log
cut -c1-400 $O/r03_bench_default.json; cut -c1-300 $O/r03_bench_k20.json
rm -rf $O/pmc/bench_* $O/pmc/loop_* $O/pmc/emd_1
