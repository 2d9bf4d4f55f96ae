# per-leg kernel stats + counter passes on the current build (one gpurun call)
bash tools/collect_legs.sh gpurun_out/legs r04 > gpurun_out/legs.log 2>&1
bash tools/collect_pmc.sh gpurun_out/pmc r04 > gpurun_out/pmc.log 2>&1
bash tools/debug/pmc_train.sh gpurun_out/pmc_train > gpurun_out/pmc_train.log 2>&1
tail -qn 3 gpurun_out/legs.log gpurun_out/pmc.log gpurun_out/pmc_train.log
