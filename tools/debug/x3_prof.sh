#!/bin/bash
# kernel-trace of the encoder forward alone at a few batch sizes: tools/debug/x3_prof.sh OUTDIR "4 8 16 32"
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for B in $1; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/b$B -- python3 $GRAFT_REPO_ROOT/tools/debug/x3_time.py $B 200 > /dev/null 2>&1 < /dev/null
  f=$(find $GRAFT_REPO_ROOT/$out/b$B -name "*kernel_stats.csv" | head -1)
  echo "B=$B"
  [ -n "$f" ] && grep -E "encoder_fwd|latent_decode" "$f" < /dev/null | cut -d, -f1-5 | cut -c1-150
done
