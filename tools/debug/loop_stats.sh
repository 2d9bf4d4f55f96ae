#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/attack_breakdown.py for the given batch sizes; prints the top kernels and copies the
# stats CSV to gpurun_out/loop_b<B>_kernel_stats.csv.   usage (on the GPU box): tools/debug/loop_stats.sh 32 [4 ...]
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rm -rf /tmp/ls_$B
  rocprofv3 --kernel-trace --stats -d /tmp/ls_$B -o ls --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/attack_breakdown.py $B > /tmp/ls_$B.log 2>&1
  f=$(find /tmp/ls_$B -name "*kernel_stats*" | head -1)
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/loop_b${B}_kernel_stats.csv
  echo "== B=$B"; grep '^{' /tmp/ls_$B.log | tail -1; head -14 "$f" | cut -d, -f1-4 | cut -c1-110
done
