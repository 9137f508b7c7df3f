#!/bin/bash
# tools/ab_dot.sh "<domain_bench.py args>" lib1 lib2 ... : same-box A/B of dot_inv_kernel builds (tools/build_tu_variant.sh),
# three alternating rounds; "tree" = the working tree's library
args=$1; shift
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for l in "$@"; do
    if [ $l = tree ]; then unset NTT_LIB; else export NTT_LIB=build/libntt_$l.so; fi
    echo "== $l round $r"; python3 tools/domain_bench.py $args | grep -v "^logn\|fwd(a),"
  done
done
