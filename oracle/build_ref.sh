#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds oracle/_ref/libgeoadv_ref.so from the reference sources
# WHERE THEY LIE under /root/reference (nothing from the reference is copied into this repo;
# oracle/_ref/ is git-ignored and holds binaries only).
#
# What is (and is not) buildable here -- see DESIGN.md "Oracle":
#   * external/grouping/test/selection_sort.cpp, query_ball_point.cpp : plain C++ files, compiled
#     WHOLE and unmodified (main renamed with -Dmain=... so the file can live in a shared object).
#   * external/structural_losses/tf_nndistance.cpp, tf_approxmatch.cpp : include TensorFlow headers
#     that this image lacks => the FILES are unbuildable.  Their CPU arithmetic, however, lives in
#     plain functions without any TF symbol; those line ranges are streamed from the file straight
#     into g++ (stdin), with the flags of tf_nndistance_compile.sh:9 (-std=c++11 -O2).  No stand-in
#     header is written: the only includes are the system headers the file itself names (:3-5).
#   * *.cu files and approxmatch.cpp (cuda_runtime.h): unbuildable (no CUDA) -- not attempted.
set -euo pipefail
REF=${GEOADV_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF/external/structural_losses" ]; then
  echo "[build_ref] $REF not present: keeping prebuilt oracle/_ref (if any)"; exit 0
fi
mkdir -p "$OUT"
TMP=$(mktemp -d /tmp/geoadv_ref.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
CXX="g++ -std=c++11 -O2 -fPIC"
NN=$REF/external/structural_losses/tf_nndistance.cpp
AM=$REF/external/structural_losses/tf_approxmatch.cpp
GT=$REF/external/grouping/test

# (1) nnsearch (tf_nndistance.cpp:21-43).  -Dstatic= gives it external linkage.
sed -n '21,43p' "$NN" | $CXX -Dstatic= -x c++ -c - -o "$TMP/nnsearch.o"

# (2) the NnDistanceGrad CPU loops (tf_nndistance.cpp:126-163) are the body of Compute(); the
#     signature line below is ours, the body is streamed from the reference.
{ echo 'void nndistance_grad_cpu(int b,int n,int m,const float*xyz1,const float*xyz2,const float*grad_dist1,const int*idx1,const float*grad_dist2,const int*idx2,float*grad_xyz1,float*grad_xyz2){';
  sed -n '126,163p' "$NN"; echo '}'; } | $CXX -x c++ -c - -o "$TMP/nngrad.o"

# (3) approxmatch_cpu / matchcost_cpu / matchcostgrad_cpu (tf_approxmatch.cpp:23-140) + its own
#     system includes (:3-5).
{ sed -n '3,5p' "$AM"; sed -n '23,140p' "$AM"; } | $CXX -x c++ -c - -o "$TMP/approxmatch.o"

# (4) whole files from external/grouping/test (they printf a lot; callers silence stdout).
$CXX -Dmain=selection_sort_main -c "$GT/selection_sort.cpp" -o "$TMP/selsort.o"
$CXX -Dmain=query_ball_point_main -c "$GT/query_ball_point.cpp" -o "$TMP/qbp.o"
# both files define randomf()/get_time(); keep one copy
objcopy --localize-symbol=_Z7randomfv "$TMP/qbp.o"

g++ -shared -o "$OUT/libgeoadv_ref.so" "$TMP/nnsearch.o" "$TMP/nngrad.o" "$TMP/approxmatch.o" "$TMP/selsort.o" "$TMP/qbp.o"
echo "[build_ref] built $OUT/libgeoadv_ref.so"
nm -D --defined-only "$OUT/libgeoadv_ref.so" | grep -E ' T ' | awk '{print "   ", $3}'
