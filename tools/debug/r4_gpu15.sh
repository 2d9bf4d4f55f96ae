cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kind in blob uniform; do for B in 32 128; do
  rm -rf gpurun_out/r4h/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4h/tr -- python3 tools/debug/emd_trace.py run $B $kind > /dev/null 2>&1
  echo "== $kind B=$B"; python3 tools/debug/emd_trace.py show gpurun_out/r4h/tr | head -9 | awk '{printf "%s %s | ", $1, $(NF-1)} END {print ""}'
done; done
