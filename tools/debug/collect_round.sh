set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; mkdir -p $O
python tools/attack_sweep.py > $O/r03_attack_sweep.json 2> $O/sweep.log
python tools/train_bench.py --steps 200 > $O/r03_train_bench.json 2>&1
python tools/train_bench.py --steps 100 --batch 50 > /dev/null 2>&1
for B in 1 4 8 16 32; do bash tools/debug/ab_cmd.sh python tools/debug/iter_timeline.py $B | grep -v "^====" > $O/r03_timeline_b$B.jsonl; done
bash tools/debug/ab_cmd.sh python tools/debug/train_timeline.py | grep -v "^====" > $O/r03_train_timeline.jsonl
python tools/debug/small_batch_paths.py 1 2 4 5 6 8 > $O/r03_small_batch_paths.jsonl
python tools/debug/hbm_roof.py > $O/r03_hbm_roof.json
tail -3 $O/sweep.log; cat $O/r03_train_bench.json | cut -c1-160
